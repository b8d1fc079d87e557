"""Data-parallel training over the GPUs of one node.

The reference is single-device (/root/reference/train.py:75-79) and issues no collective.  A batch
is a disjoint union of graphs (block-diagonal adjacency, graphs contiguous -- what PyG's collate
builds for /root/reference/train.py:108-109), and neither the forward nor the backward of
/root/reference/model.py:26-45 exchanges anything between graphs, so the path shards naturally:

* one process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on ROCm, ``gloo`` for the
  CPU tests), identical parameter replicas, identical Adam state;
* rank r trains on a contiguous range of graphs (``shard_batch``: balanced by nodes+edges, not by graph
  count, because COLLAB-like degree skew makes graphs very unequal);
* exactly ONE collective per step: all-reduce(sum) of the flat fp32 gradient buffer (~52 k floats =
  208 KB, latency-bound: ring wire time ~2.4 us at 153 GB/s per xGMI link, so a single flat bucket,
  never per-parameter messages);
* the loss is the MEAN over the GLOBAL batch (``nn.NLLLoss()`` default, train.py:98), so every rank
  scales its label gradients by 1/B_global; sum-all-reduce then reproduces the 1-GPU gradient for the
  same global batch (up to fp32 summation order).

``GradAllReduce`` is engine-agnostic: it works on any flat tensor, so the same code is exercised on CPU
(gloo, with the oracle as compute in tests/) and on GPU (RCCL, with the HIP path).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .batch import Batch, split_batch


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from torchrun's environment.
    Returns (rank, world_size, local_rank).  Rendezvous on 127.0.0.1 unless MASTER_ADDR says otherwise."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(batch: Batch, rank: int, world_size: int) -> Batch:
    """Rank ``rank``'s contiguous, cost-balanced range of graphs of a (host-resident) global batch."""
    if world_size == 1:
        return batch
    return split_batch(batch, world_size)[rank]


def graph_range(num_graphs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[g0, g1) of an even split by graph count (used when every rank generates its own graphs)."""
    base, rem = divmod(num_graphs, world_size)
    g0 = rank * base + min(rank, rem)
    return g0, g0 + base + (1 if rank < rem else 0)


class GradAllReduce:
    """One flat-bucket gradient all-reduce per step."""

    def __init__(self, process_group=None, force: bool = False):
        """``force``: issue the collective even in a 1-rank group (lets a 1-GPU box time the N>1 code path)."""
        self.pg = process_group
        self.force = bool(force)
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1

    def __call__(self, flat_grad: torch.Tensor, async_op: bool = False):
        if self.world_size == 1 and not self.force:
            return None
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)

    def global_batch(self, local_graphs: int, device=None) -> int:
        """Sum of the ranks' local batch sizes (one tiny all-reduce; call once per epoch/config, not per step,
        when the split is static)."""
        if self.world_size == 1:
            return int(local_graphs)
        t = torch.tensor([local_graphs], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return int(t.item())


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, process_group=None) -> None:
    """Make every replica identical to rank ``src`` (one broadcast of the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_params, src=src, group=process_group)


def flatten_grads(params: Sequence[torch.nn.Parameter]) -> Tuple[torch.Tensor, List[Tuple[int, int, torch.Size]]]:
    """Pack ``p.grad`` of arbitrary parameters into one flat tensor (for engines without a native flat
    buffer, e.g. the CPU oracle in the tests).  Returns (flat, [(offset, numel, shape)...])."""
    metas, chunks, off = [], [], 0
    for p in params:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        chunks.append(g.reshape(-1))
        metas.append((off, g.numel(), g.shape))
        off += g.numel()
    return torch.cat(chunks), metas


def unflatten_grads(flat: torch.Tensor, metas, params: Sequence[torch.nn.Parameter]) -> None:
    for p, (off, n, shape) in zip(params, metas):
        p.grad = flat[off:off + n].view(shape).clone()


class _RawCudaBuffer:
    """``__cuda_array_interface__`` over foreign device memory, so that torch can view it (no ownership)"""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def _host_id() -> str:
    import socket
    return socket.gethostname()


def _device_identity(device: torch.device) -> str:
    """physical identity of ``device``: uuid AND PCI address together -- NOT the local ordinal: ranks launched with their own
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES all see index 0, and different GPUs must not look like one device to the
    same-device test that admits coarse-grained exchange memory.  A uuid alone is not trusted: builds that report an all-zero
    (or otherwise constant) uuid for every GPU would make distinct devices compare equal, so an all-zero uuid is dropped and
    the PCI domain:bus:device is always part of the identity when the driver reports it; with neither, every rank counts as a
    device of its own (coarse-grained memory is then refused across ranks)."""
    import os
    parts = []
    try:
        props = torch.cuda.get_device_properties(device)
        u = getattr(props, "uuid", None)
        if u not in (None, "", 0):
            us = str(u)
            if any(ch not in "0-{} " for ch in us):          # something other than zeros and separators
                parts.append(f"uuid:{us}")
        bus = getattr(props, "pci_bus_id", None)
        if bus is not None and not (bus == 0 and getattr(props, "pci_device_id", 0) == 0 and not parts):
            parts.append(f"pci:{getattr(props, 'pci_domain_id', 0)}:{bus}:{getattr(props, 'pci_device_id', 0)}")
    except Exception:
        parts = []
    if not parts:
        return f"unknown:{_host_id()}:{os.getpid()}"
    return "|".join(parts)


class PeerExchange:
    """One-shot gradient all-reduce + Adam over peer-mapped memory (``dgcnn_allreduce_adam_step``, csrc/peer.hip).

    Every rank allocates [flag block 64 B | gradient buffer 0 | gradient buffer 1] in fine-grained device memory, the
    ranks swap the IPC handles through the process group (any backend: the handles are 64 opaque bytes) and map each
    other's blocks -- peers on other GPUs of the node over xGMI, or, as in the tests of this repository (one GPU), another
    process on the same device.  The weight-gradient kernel writes straight into the current buffer; ONE launch per rank
    then replaces ``all_reduce`` + the optimizer launch.  No multi-GPU hardware was available to time it:
    it is opt-in (``Trainer(..., one_shot=True)``), RCCL stays the default.
    """

    HDR = 64

    def __init__(self, numel: int, process_group=None, device=None):
        import ctypes
        from . import _lib
        self._lib = _lib
        L = _lib.lib()
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.numel = int(numel)
        self.stride = ((self.numel * 4 + 255) // 256) * 256
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        nbytes = self.HDR + 2 * self.stride
        self._own, self._opened, self._bases = None, [], []
        # set-up is collective but its failures are local: every phase ends with an exchange of success flags, and a
        # failure on ANY rank releases what was mapped and raises on EVERY rank (nobody is left waiting in a barrier)
        own = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        fail = None
        rc = L.dgcnn_peer_alloc(nbytes, ctypes.byref(own), handle)
        if rc != 0:
            fail = f"dgcnn_peer_alloc failed ({rc}) on rank {self.rank}"
        else:
            self._own = own.value
            self.fine_grained = bool(L.dgcnn_peer_last_alloc_finegrained())
        infos = [None] * self.world
        mine = (bytes(handle.raw), fail, _device_identity(self.device), bool(getattr(self, "fine_grained", False)), _host_id())
        if self.world > 1:
            dist.all_gather_object(infos, mine, group=process_group)
        else:
            infos[0] = mine
        fails = [i[1] for i in infos if i[1]]
        # coarse-grained exchange memory is only coherent for flag polling between processes of ONE device
        if not fails and len({(i[4], i[2]) for i in infos}) > 1 and not all(i[3] for i in infos):
            fails.append("fine-grained device memory is unavailable on a rank and the ranks span more than one device: "
                         "the one-shot exchange cannot poll its flags coherently (use the collective route)")
        if not fails:
            for r in range(self.world):
                if r == self.rank:
                    self._bases.append(self._own)
                    continue
                p = ctypes.c_void_p()
                hb = ctypes.create_string_buffer(infos[r][0], 64)
                rc = L.dgcnn_peer_open(hb, ctypes.byref(p))
                if rc != 0:
                    fail = f"dgcnn_peer_open(rank {r}) failed ({rc}) on rank {self.rank}"
                    break
                self._bases.append(p.value)
                self._opened.append(p.value)
            oks = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(oks, fail, group=process_group)
            else:
                oks[0] = fail
            fails = [f for f in oks if f]
        if fails:
            self.release_local()
            raise _lib.DgcnnError("one-shot exchange set-up failed: " + "; ".join(fails))
        arr = ctypes.c_void_p * self.world
        self._flags = arr(*[b for b in self._bases])
        self._grads = [arr(*[b + self.HDR + par * self.stride for b in self._bases]) for par in (0, 1)]
        self.err = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._views = [torch.as_tensor(_RawCudaBuffer(self._own + self.HDR + par * self.stride, self.numel),
                                       device=self.device) for par in (0, 1)]
        if self.world > 1:
            dist.barrier(group=process_group)          # everybody has mapped everybody before the first step

    def grad_ptr(self, parity: int) -> int:
        return self._own + self.HDR + (parity & 1) * self.stride

    def grad_tensor(self, parity: int) -> torch.Tensor:
        return self._views[parity & 1]

    def step(self, tag: int, params: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, adam_step: int,
             lr: float, betas, eps: float, stream: int, grad_sum_out: Optional[torch.Tensor] = None) -> None:
        """sum of every rank's buffer ``tag & 1`` in rank order + Adam on this replica; ``tag`` = 1, 2, 3, ..."""
        if getattr(self, "_aborted", False):
            raise self._lib.DgcnnError("one-shot all-reduce: an earlier step was aborted (lost or stalled rank); the replicas "
                                       "may have diverged by that step -- no further step is issued")
        rc = self._lib.lib().dgcnn_allreduce_adam_step(
            self.world, self.rank, self._grads[tag & 1], self._flags, tag, params.data_ptr(), exp_avg.data_ptr(),
            exp_avg_sq.data_ptr(), None if grad_sum_out is None else grad_sum_out.data_ptr(), self.numel, adam_step,
            lr, betas[0], betas[1], eps, self.err.data_ptr(), stream)
        self._lib.check(rc, "dgcnn_allreduce_adam_step")

    def check(self) -> None:
        """host-side check (a sync): did a step of the exchange time out?  The kernel's verdict is agreed by all ranks (a
        step any rank gave up on is applied by none), so every rank raises here for the same step."""
        if getattr(self, "_aborted", False) or int(self.err[0].item()) != 0:
            self._aborted = True          # sticky, like the kernel's word: no further step is issued (step() refuses)
            raise self._lib.DgcnnError("one-shot all-reduce: a peer did not publish its gradient within the bounded wait "
                                       "(lost or stalled rank); the step was applied on no rank unless a peer's verdict itself was lost, and "
                                       "no further step is applied anywhere "
                                       "(dgcnn_peer_set_timeout_ms raises the bound)")

    def release_local(self) -> None:
        """unmap the peers and free the own block WITHOUT any collective (error paths, ``__del__``)"""
        L = self._lib.lib()
        for p in getattr(self, "_opened", []):
            L.dgcnn_peer_close(p)
        if getattr(self, "_own", None) is not None:
            L.dgcnn_peer_free(self._own)
        self._own, self._opened = None, []

    def __del__(self):
        try:
            self.release_local()
        except Exception:
            pass

    def close(self) -> None:
        L = self._lib.lib()
        if getattr(self, "_own", None) is None:
            return
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier(group=self.pg)                # nobody still reads a buffer that is about to go away
        self.release_local()
