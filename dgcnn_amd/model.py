"""``Model`` -- drop-in replacement of the reference's network class, running on the
hand-written HIP path behind the C ABI (``include/dgcnn_hip.h``).

Boundary kept from the reference:

* ``Model(num_features, num_classes)``                     /root/reference/model.py:10
* ``forward(data) -> [B, num_classes]`` log-probabilities, reading exactly ``data.x``,
  ``data.edge_index``, ``data.batch``                       /root/reference/model.py:26-45
* an ``nn.Module``: ``.to(device)``, ``.parameters()``, ``.train()/.eval()`` (Dropout),
  ``.state_dict()`` with the reference's key names (PyG GCNConv: ``convN.lin.weight``,
  ``convN.bias``; ``conv5/conv6/classifier_1/classifier_2`` ``.weight/.bias``)
  /root/reference/train.py:97-99,129
* differentiable wrt every parameter (``loss.backward()``, train.py:40).

Inside, nothing of the reference's op sequence is replayed with torch ops: forward is one
``dgcnn_model_forward`` call, backward one ``dgcnn_model_backward`` call, each a chain of HIP
kernel launches on the current torch stream.  All parameters live in ONE flat fp32 buffer
(``flat_params``); the ``nn.Parameter`` objects are views into it, so the same buffer is the
gradient all-reduce bucket and the fused-Adam operand.

There is NO CPU/eager fallback: on a CPU tensor, or without ``libdgcnn_hip.so``, this raises.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch import nn

from . import _lib

K_SORT = 30
_NO_EDGES = torch.zeros(2, 0, dtype=torch.int64)      # stands in for edge_index of a PreparedBatch (never dereferenced)


class _Lin(nn.Module):
    """Holds ``weight`` so the key is ``convN.lin.weight`` like PyG's GCNConv."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))


class GCNConvParams(nn.Module):
    """Parameter container of one graph-convolution layer (``GCNConv(fin, fout)`` at
    /root/reference/model.py:13-16).  PyG init: glorot-uniform weight, zero bias."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.in_channels, self.out_channels = fin, fout
        self.lin = _Lin(fin, fout)
        self.bias = nn.Parameter(torch.zeros(fout))
        a = math.sqrt(6.0 / (fin + fout))
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}"


class SortPoolSpec(nn.Module):
    """Stands where ``SortAggregation(k=30)`` stands in the reference (model.py:17)."""

    def __init__(self, k: int = K_SORT):
        super().__init__()
        self.k = k

    def extra_repr(self) -> str:
        return f"k={self.k}"


def _batch_size_of(data) -> int:
    B = getattr(data, "num_graphs", None)
    if B is None:
        B = int(data.batch[-1].item()) + 1      # host sync, as in the reference's SortAggregation
    return int(B)


_WS_BYTES = {}      # (N, E, B, F, C) -> dgcnn_workspace_bytes: a pure function of the dimensions, memoised (one ctypes call per step otherwise)


class _DGCNNFunction(torch.autograd.Function):
    """forward = dgcnn_model_forward, backward = dgcnn_model_backward (one C call each)."""

    @staticmethod
    def forward(ctx, model, x, edge_index, batch, B, training, seed, flags, max_nodes, max_edges, pb, *params):
        L = _lib.lib()
        N, F = x.shape
        E = edge_index.shape[1] if pb is None else pb.num_edges
        C = model.num_classes
        flat = model.flat_params_fast()
        wkey = (N, E, B, F, C)
        nbytes = _WS_BYTES.get(wkey)
        if nbytes is None:
            if len(_WS_BYTES) > 4096:
                _WS_BYTES.clear()
            nbytes = _WS_BYTES[wkey] = _lib.workspace_bytes(N, E, B, F, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        logp = torch.empty(B, C, dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        epoch = model._next_epoch()
        if pb is not None:
            # batch of a PreparedDataset (dgcnn_amd/device_data.py): its structures are COPIED into the workspace from the
            # dataset's (one launch), then the forward runs as on a prepared workspace -- no edge list exists
            _lib.check(L.dgcnn_assemble(pb.dataset.desc_ref, B, N, E, C, pb.ids_ptr, pb.onode_ptr, pb.oedge_ptr, ws.data_ptr(),
                                        x.data_ptr(), batch.data_ptr(), pb.y.data_ptr(), flags, max_nodes, epoch, stream),
                       "dgcnn_assemble")
            flags |= _lib.FLAG_PREPARED
        _lib.check(L.dgcnn_model_forward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(),
                                         edge_index.data_ptr() if (E and pb is None) else None, batch.data_ptr(),
                                         ws.data_ptr(), logp.data_ptr(), int(training), seed, flags,
                                         max_nodes, max_edges, epoch, stream),
                   "dgcnn_model_forward")
        ctx.model = model
        ctx.dims = (N, E, B, F, C, int(training), int(flags), int(max_nodes))
        ctx.save_for_backward(x, ws, logp)
        md = model.__dict__          # (nn.Module.__setattr__ costs ~2 us per assignment)
        md["_last_ws"] = ws
        md["_last_dims"] = wkey
        return logp

    @staticmethod
    def backward(ctx, glogp):
        L = _lib.lib()
        model = ctx.model
        N, E, B, F, C, training, flags, max_nodes = ctx.dims
        x, ws, logp = ctx.saved_tensors
        glogp = glogp.contiguous()
        flat = model.flat_params_fast()
        grads = torch.zeros_like(flat)       # fresh buffer: p.grad views stay valid until dropped
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(L.dgcnn_model_backward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(), ws.data_ptr(),
                                          logp.data_ptr(), glogp.data_ptr(), None, 0.0, training,
                                          grads.data_ptr(), None, flags, max_nodes, stream), "dgcnn_model_backward")
        model.__dict__["_last_flat_grad"] = grads
        # the 16 gradients handed to autograd are views of ONE flat buffer in the parameter layout: autograd keeps them
        # as they are (no copies), so an optimizer that recognises the layout (dgcnn_amd.optim.Adam) updates the
        # whole model with one kernel
        return (None, None, None, None, None, None, None, None, None, None, None, *model._grad_views(grads))


class Model(nn.Module):
    def __init__(self, num_features: int, num_classes: int):
        super().__init__()
        self.num_features, self.num_classes = int(num_features), int(num_classes)
        # same attribute names as /root/reference/model.py:13-24 (parameter containers only;
        # none of these torch modules is ever called)
        self.conv1 = GCNConvParams(num_features, 32)
        self.conv2 = GCNConvParams(32, 32)
        self.conv3 = GCNConvParams(32, 32)
        self.conv4 = GCNConvParams(32, 1)
        self.sort_pool = SortPoolSpec(K_SORT)
        self.conv5 = nn.Conv1d(1, 16, 97, 97)
        self.conv6 = nn.Conv1d(16, 32, 5, 1)
        self.pool = nn.MaxPool1d(2, 2)
        self.classifier_1 = nn.Linear(352, 128)
        self.drop_out = nn.Dropout(0.5)
        self.classifier_2 = nn.Linear(128, num_classes)
        self.relu = nn.ReLU(inplace=True)
        self._flat: Optional[torch.Tensor] = None
        self._offsets: Optional[List[int]] = None
        self._total = 0
        self._fwd_count = 0
        self._seed_base: Optional[int] = None
        self._last_ws = None
        self._last_dims = None
        self._last_flat_grad = None
        self._epoch = 0

    # ---- flat parameter buffer ---------------------------------------------------------
    def _param_list(self) -> List[nn.Parameter]:
        """Parameters in the order of the C-ABI flat layout (include/dgcnn_hip.h).  The Parameter OBJECTS never change
        (``.to()`` / ``load_state_dict`` replace or fill their data), so the list is built once: 16 ``nn.Module``
        attribute look-ups cost ~20 us, too much for a per-step path."""
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = [self.conv1.lin.weight, self.conv1.bias, self.conv2.lin.weight, self.conv2.bias,
                  self.conv3.lin.weight, self.conv3.bias, self.conv4.lin.weight, self.conv4.bias,
                  self.conv5.weight, self.conv5.bias, self.conv6.weight, self.conv6.bias,
                  self.classifier_1.weight, self.classifier_1.bias,
                  self.classifier_2.weight, self.classifier_2.bias]
            self.__dict__["_plist"] = pl
        return pl

    def _grad_views(self, flat_grad: torch.Tensor):
        """the 16 per-parameter views of a flat gradient buffer: ONE split call + reshapes of the matrices"""
        sp = self.__dict__.get("_gsplit")
        if sp is None:
            offs, plist = self._offsets, self._param_list()
            sizes, keep, shapes, pos = [], [], [], 0
            for p, off in zip(plist, offs):
                if off > pos:
                    sizes.append(off - pos)          # alignment gap
                sizes.append(p.numel()); keep.append(len(sizes) - 1); shapes.append(tuple(p.shape))
                pos = off + p.numel()
            if self._total > pos:
                sizes.append(self._total - pos)
            sp = (sizes, keep, shapes)
            self.__dict__["_gsplit"] = sp
        sizes, keep, shapes = sp
        parts = flat_grad.split_with_sizes(sizes)
        return [parts[k] if len(sh) == 1 else parts[k].view(sh) for k, sh in zip(keep, shapes)]

    def _views_of(self, flat: torch.Tensor) -> List[torch.Tensor]:
        out = []
        for p, off in zip(self._param_list(), self._offsets):
            out.append(flat[off:off + p.numel()].view(p.shape))
        return out

    def _is_flat(self) -> bool:
        if self._flat is None:
            return False
        base = self._flat.data_ptr()
        for p, off in zip(self._param_list(), self._offsets):
            if p.data_ptr() != base + 4 * off or p.device != self._flat.device:
                return False
        return True

    def flatten_parameters(self) -> torch.Tensor:
        """(Re)pack all parameters into one flat fp32 device buffer and re-point each
        ``nn.Parameter`` at its slice.  Called lazily; needed again after ``.to()`` or
        ``load_state_dict`` moved/replaced parameter storage."""
        plist = self._param_list()
        dev = plist[0].device
        if self._offsets is None:
            self._offsets, self._total = _lib.param_layout(self.num_features, self.num_classes)
        flat = torch.zeros(self._total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(plist, self._offsets):
                if p.dtype != torch.float32:
                    raise _lib.DgcnnError("dgcnn_amd.Model computes in fp32; parameters must be float32")
                flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
            for p, off in zip(plist, self._offsets):
                p.data = flat[off:off + p.numel()].view(p.shape)
        self._flat = flat
        return flat

    @property
    def flat_params(self) -> torch.Tensor:
        if not self._is_flat():
            self.flatten_parameters()
        return self._flat

    def _apply(self, fn, *args, **kwargs):
        """``.to()`` / ``.cuda()`` / ``.float()`` replace parameter storage: drop the flat buffer (rebuilt lazily)."""
        self._flat = None
        self.__dict__.pop("_gsplit", None)
        return super()._apply(fn, *args, **kwargs)

    def flat_params_fast(self) -> torch.Tensor:
        """``flat_params`` for the per-step hot path: instead of checking all 16 parameters (~15 us of Python per
        call) it probes the first and the last one; ``_apply`` (``.to()`` & co) and ``load_state_dict`` are covered
        exactly (``_apply`` drops the buffer, ``load_state_dict`` copies in place).  Replacing a single
        ``param.data`` by hand between steps needs a ``flatten_parameters()`` call."""
        flat = self._flat
        if flat is not None:
            base = flat.data_ptr()
            offs = self._offsets
            if self.conv1.lin.weight.data_ptr() == base + 4 * offs[0] and \
                    self.classifier_2.bias.data_ptr() == base + 4 * offs[15]:
                return flat
        return self.flat_params

    @property
    def flat_numel(self) -> int:
        if self._offsets is None:
            self._offsets, self._total = _lib.param_layout(self.num_features, self.num_classes)
        return self._total

    # ---- forward -----------------------------------------------------------------------
    # (per-step bookkeeping goes through __dict__: nn.Module.__setattr__ costs ~1.5 us per assignment)
    def _next_seed(self) -> int:
        d = self.__dict__
        if d["_seed_base"] is None:
            d["_seed_base"] = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        d["_fwd_count"] += 1
        return (d["_seed_base"] * 0x9E3779B97F4A7C15 + d["_fwd_count"] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _next_epoch(self) -> int:
        """non-zero 32-bit tag of a forward call (error words in the workspace are epoch-tagged)."""
        d = self.__dict__
        e = d["_epoch"] = (d["_epoch"] % 0x7FFFFFFE) + 1
        return e

    def _mode_flags(self) -> int:
        """path overrides set as plain attributes (tests, sweeps, bench.py); every one defaults to the library's choice:
        ``use_fused`` True/False, ``agg_mode`` "dense"/"sparse", ``use_chain`` True/False (graph-chain kernels),
        ``compute_dtype`` "bf16" (BASELINE config 3's leg)."""
        d = self.__dict__
        f = 0
        uf = d.get("use_fused")
        if uf is True:
            f |= _lib.FLAG_FORCE_FUSED
        elif uf is False:
            f |= _lib.FLAG_FORCE_TILED
        am = d.get("agg_mode")
        if am == "dense":
            f |= _lib.FLAG_AGG_DENSE
        elif am == "sparse":
            f |= _lib.FLAG_AGG_SPARSE
        if d.get("compute_dtype") == "bf16":
            f |= _lib.FLAG_BF16
        uc = d.get("use_chain")
        if uc is True:
            f |= _lib.FLAG_CHAIN
        elif uc is False:
            f |= _lib.FLAG_NO_CHAIN
        return f

    def _flags_of(self, data) -> int:
        f = _lib.FLAG_COALESCED_UNDIRECTED if getattr(data, "coalesced_undirected", False) else 0
        # ``mode_flags``: a form restriction that belongs to the DATA (a PreparedDataset built without bitmap rows hands out
        # batches that must stay on the CSR kernels: device_data.PreparedBatch)
        f |= int(getattr(data, "mode_flags", 0) or 0) | self._mode_flags()
        if not torch.is_grad_enabled():
            f |= self._inference_flag()
        return f

    def _inference_flag(self) -> int:
        """``DGCNN_FLAG_INFERENCE`` for a forward no backward can follow (``torch.no_grad()``, ``Trainer.eval_step``) when the model
        attribute ``inference_one_launch`` is set: small batches with a graph of 257..512 nodes then take the one-launch evaluation
        kernel too (round 6: ``test()`` of the reference, train.py:49-66, one launch per batch on PROTEINS-like sets).  Verified on
        the CPU emulation; OFF by default because it is not expected to be faster: the launch lasts as long as its largest graph,
        whose block products grow with n^2 (round 5 measured the two-tiles-per-wave chain forward of such batches ~13 us SLOWER
        than the four gather launches it replaces) -- `tools/eval_route_time.py` decides on the first GPU call."""
        return _lib.FLAG_INFERENCE if self.__dict__.get("inference_one_launch") else 0

    def _max_nodes_of(self, data) -> int:
        """per-graph node bound (host-known hint) for the graph-per-workgroup path"""
        return int(getattr(data, "max_nodes", 0) or 0)

    def check_errors(self, workspaces=None, since: int = -1) -> None:
        """Host-side check (one tiny D2H copy per workspace = a sync) of the input-error words the forward calls left
        behind.  The kernels never mis-compute silently: an out-of-range edge endpoint or a violated
        ``coalesced_undirected`` promise is flagged here.

        Default: the most recent forward's workspace and tag only.  ``workspaces`` (list of (ws, dims)) with ``since``
        (an earlier value of the forward counter): any error a forward with tag in (since, now] left in ANY of those
        workspaces -- the words are epoch-tagged and only ever overwritten by a later error, so one check per epoch of
        training covers every batch of it (``Trainer.read_metrics``).  Drop-in users of ``Model.forward`` who
        want the guarantee call ``check_errors()`` after the batches they care about."""
        if workspaces is None:
            if self._last_ws is None:
                return
            workspaces, since = [(self._last_ws, self._last_dims)], self._epoch - 1
        now = self._epoch
        for ws, dims in workspaces:
            if ws is None:
                continue
            self._check_err_words(_lib.ws_view(ws, "err", *dims).cpu().tolist(), since, now)

    @staticmethod
    def _check_err_words(words, since: int, now: int) -> None:
        """the four epoch-tagged error words of ONE workspace, already on the host (``Trainer.read_metrics`` fetches the words
        of both of its workspace slots together with the metrics in a single device-to-host copy)"""
        u = [v & 0xFFFFFFFF for v in words]
        for k, msg in ((0, "edge_index holds a node id outside [0, N)"),
                       (1, "a host-side promise about the batch does not hold: coalesced_undirected "
                           "(edge_index sorted by (src,dst), no duplicates/self loops, reverse edges "
                           "present), max_nodes (too small), or block-diagonality (an edge leaves its "
                           "graph) -- the forward result of this batch is invalid"),
                       (4, "an in-launch wait of the pipelined graph preparation timed out: exclusive_device was promised "
                           "but something else (a second process, another stream's waiting kernels) held the compute units "
                           "-- the batch's structures are incomplete; use Trainer(..., exclusive_device=False)")):
            if k + 2 >= len(u):
                continue
            e = u[k]
            if e != 0 and u[k + 2] == ((~e) & 0xFFFFFFFF) and (since < e <= now or (now < since and (e > since or e <= now))):
                raise _lib.DgcnnError(msg)

    @staticmethod
    def _check_inputs(x, edge_index, batch):
        if not x.is_cuda:
            raise _lib.DgcnnError(
                "dgcnn_amd.Model runs only on an AMD GPU through libdgcnn_hip.so; got a CPU tensor "
                "(there is deliberately no CPU fallback)")
        if x.dtype != torch.float32 or x.dim() != 2:
            raise _lib.DgcnnError(f"data.x must be [N,F] float32, got {x.dtype} {tuple(x.shape)}")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise _lib.DgcnnError("data.edge_index must be [2,E] int64")
        if batch.dtype != torch.int64 or batch.shape[0] != x.shape[0]:
            raise _lib.DgcnnError("data.batch must be [N] int64")

    def forward(self, data):
        x, edge_index, batch = data.x, data.edge_index, data.batch          # model.py:27
        pb = data if getattr(data, "dataset", None) is not None else None    # PreparedBatch: no edge list, structures per dataset
        if pb is not None:
            edge_index = _NO_EDGES
            if not x.is_cuda or batch.shape[0] != x.shape[0]:
                raise _lib.DgcnnError("PreparedBatch buffers must be CUDA tensors [N,F] / [N]")
        else:
            self._check_inputs(x, edge_index, batch)
        if x.shape[1] != self.num_features:
            raise _lib.DgcnnError(f"data.x has {x.shape[1]} features, model expects {self.num_features}")
        if pb is None:
            x, edge_index, batch = x.contiguous(), edge_index.contiguous(), batch.contiguous()
        B = _batch_size_of(data)
        flat = self.flat_params_fast()
        if flat.device != x.device:
            raise _lib.DgcnnError(f"model on {flat.device}, data on {x.device}")
        training = self.training
        seed = self._next_seed() if training else 0
        return _DGCNNFunction.apply(self, x, edge_index, batch, B, training, seed, self._flags_of(data),
                                    self._max_nodes_of(data), int(getattr(data, "max_edges", 0) or 0), pb,
                                    *self._param_list())

    # ---- introspection used by tests / tools ---------------------------------------------
    def last_workspace_view(self, name: str) -> torch.Tensor:
        """Typed view of a region of the workspace of the most recent forward."""
        if self._last_ws is None:
            raise _lib.DgcnnError("no forward has run yet")
        return _lib.ws_view(self._last_ws, name, *self._last_dims)
