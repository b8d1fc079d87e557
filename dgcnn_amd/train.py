"""Training / evaluation step harness on the fused HIP path.

Restates the per-batch body of the reference loops -- ``train()`` /root/reference/train.py:27-47
and ``test()`` train.py:49-66 -- with the same semantics:

    pred = model(data); loss = NLLLoss()(pred, y); loss.backward()
    optimizer.step(); optimizer.zero_grad()                       (Adam defaults, train.py:99)
    running_loss += loss.item(); correct += (pred.argmax(1) == y).sum().item()

but as four C-ABI calls per batch and NO host synchronisation: forward, backward (NLL gradient
generated in-kernel from the labels), [one flat gradient all-reduce when data-parallel], fused
Adam + zero_grad; loss / #correct accumulate in a 2-float device buffer that is read once per
epoch (the reference syncs twice per batch, train.py:44-45).

PyG / visdom cannot travel to the GPU box, so the reference's ``train.py`` itself cannot run there;
this is the build's own harness for the same step.  The drop-in route (reference loop + this
build's ``Model`` + torch's Adam) also works and is what tests/test_gpu_model.py
(``test_fused_train_step_equals_dropin_route_and_oracle_loss``, ``test_dropin_optimizer_*``) exercises.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import math

import torch

from . import _lib
from .model import Model, _batch_size_of


class Trainer:
    """Fused train/eval step for a :class:`dgcnn_amd.Model`.

    Batches are recognised by object identity: a loop that re-uses resident ``Batch`` objects (bench.py's pool, a
    dataset that fits on the device) pays the argument-block setup once per batch object; up to ``ARGS_CACHE_MAX``
    batches are remembered (each entry keeps that batch's tensors alive, so the bound is also a memory bound).

    lr/betas/eps default to ``torch.optim.Adam`` defaults, which is what the reference uses
    (``Adam(model.parameters())``, /root/reference/train.py:99).
    ``process_group``: when given (data parallel, one process per GPU), gradients are summed over
    ranks with ONE all-reduce of the flat buffer per step and the loss is scaled by the GLOBAL
    batch size, so N ranks x B/N graphs reproduce one rank x B graphs.
    """

    ARGS_CACHE_MAX = 128

    def __init__(self, model: Model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, force_collective: bool = False, one_shot: bool = False,
                 exclusive_device: Optional[bool] = None):
        """``exclusive_device``: the promise behind ``DGCNN_FLAG_EXCLUSIVE_DEVICE`` -- nothing else runs on this GPU while a
        step is in flight -- which admits the form of a small batch's step whose launch ALSO carries both phases of the next
        batch's graph preparation (workgroups of one launch waiting for each other on the device).  A process cannot see what
        other processes run on its GPU, so the default (``None``) is False: the promise is the CALLER's to make
        (``bench.py`` and the measurement tools pass True where every rank owns its GPU, ``python train.py
        --exclusive-device`` does).  Without it the next batch's phase B rides on ``k_wgrad`` -- no
        workgroup ever waits for another, whatever else runs on the device."""
        self.model = model
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.pg = process_group
        from .dist import GradAllReduce
        self._allreduce = GradAllReduce(process_group, force=force_collective) if process_group is not None else None
        if self._allreduce is not None and self._allreduce.world_size == 1 and not force_collective:
            self._allreduce = None           # a 1-rank group needs no collective
        self._dp_world = self._allreduce.world_size if self._allreduce is not None else 1
        if process_group is not None and self._allreduce is None:
            import torch.distributed as dist
            self._dp_world = dist.get_world_size(process_group)
        self._err_checked = model._epoch      # forward tag up to which input errors have been surfaced
        if exclusive_device is None:
            exclusive_device = False
        self._excl = _lib.FLAG_EXCLUSIVE_DEVICE if exclusive_device else 0
        # one-shot exchange (dgcnn_amd.dist.PeerExchange): gradients land in peer-mapped memory, ONE kernel per rank sums
        # them in rank order and applies Adam -- instead of all_reduce + dgcnn_adam_step.  Opt-in (no multi-GPU timing yet).
        self._one_shot = bool(one_shot) and self._allreduce is not None
        self._peer = None
        self.step_count = 0
        flat = model.flat_params
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.grads = torch.zeros_like(flat)
        self.metrics = torch.zeros(2, dtype=torch.float32, device=flat.device)
        self._logp = None
        # two workspace slots: the step runs in slot `_cur` while the pipelined step prepares the NEXT batch's graph
        # structure in the other slot on the library's side stream (software pipelining of graph prep across steps)
        self._slots = [{"ws": None, "bytes": 0, "ptr": 0}, {"ws": None, "bytes": 0, "ptr": 0}]
        self._cur = 0
        self._pipe = None        # dgcnn_pipeline handle, created on first pipelined step
        self._pipe_fn = None
        self._pipe_eval_fn = None
        self._p_flat = 0         # cached data_ptr()s of the trainer-lifetime buffers (pipelined step)
        self._p_grads = self._p_metrics = self._p_m = self._p_v = 0
        self._logp_views = {}
        self._args_cache = {}    # id(batch) -> (batch, y, StepArgs, ws bytes, keep-alive tensors, dims)
        self._prep_ent = None    # cache entry of the batch whose graph structure the last pipelined call prepared
        self._prep_slot = 0

    def close(self) -> None:
        """release the peer-mapped exchange block (collective: every rank calls it)"""
        if self._peer is not None:
            self._peer.close()
            self._peer = None

    def __del__(self):
        try:
            if self._pipe is not None:
                _lib.lib().dgcnn_pipeline_destroy(self._pipe)
                self._pipe = None
            if self._peer is not None:           # local unmap / free only: a destructor must not enter a collective
                self._peer.release_local()
                self._peer = None
        except Exception:
            pass

    # ---- buffers reused across steps (sizes only grow) -------------------------------------
    def _slot_ws(self, k, need, device):
        sl = self._slots[k]
        if sl["ws"] is None or need > sl["bytes"] or sl["ws"].device != device:
            if sl["ws"] is not None and sl.get("dims"):
                # the buffer about to be dropped carries the sticky error words of every batch it served since the last
                # check: surface them now (one small copy; growth is rare) instead of losing them with the buffer
                self.model.check_errors([(sl["ws"], sl["dims"])], since=self._err_checked)
            sl["ws"] = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=device)
            sl["ws"][:32].zero_()          # error words of a fresh buffer: no stale tag can pass for a real one
            sl["bytes"] = sl["ws"].numel()
            sl["ptr"] = sl["ws"].data_ptr()
        return sl["ws"]

    def _buffers(self, N, E, B, F, C, device):
        need = _lib.workspace_bytes(N, E, B, F, C)
        if self._prep_ent is not None:       # an unpipelined call interleaved: keep the prepared slot intact
            self._cur = 1 - self._prep_slot
        self._ws = self._slot_ws(self._cur, need, device)
        self._slots[self._cur]["dims"] = (N, E, B, F, C)
        if self._logp is None or self._logp.shape[0] < B or self._logp.shape[1] != C or self._logp.device != device:
            self._logp = torch.empty(max(B, 64), C, dtype=torch.float32, device=device)
        return self._ws, self._logp

    def _dims(self, data):
        x, ei = data.x, data.edge_index
        if getattr(data, "dataset", None) is not None:       # PreparedBatch (device_data.py): sizes are host-known, no edge list
            return data.num_nodes, data.num_edges, data.num_graphs, x.shape[1], self.model.num_classes
        Model._check_inputs(x, ei, data.batch)
        return x.shape[0], ei.shape[1], _batch_size_of(data), x.shape[1], self.model.num_classes

    def forward_backward(self, data, y, global_batch: Optional[int] = None, fuse_adam: bool = False) -> torch.Tensor:
        """forward + NLL(mean) + backward into ``self.grads``; returns the log-probs view [B,C].
        No sync.  ``fuse_adam``: the weight-gradient kernel also applies the Adam update
        (``dgcnn_model_backward_step``) -- single-GPU only."""
        L = _lib.lib()
        m = self.model
        N, E, B, F, C = self._dims(data)
        dev = data.x.device
        ws, logp = self._buffers(N, E, B, F, C, dev)
        flat = m.flat_params
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        training = 1 if m.training else 0
        seed = m._next_seed() if training else 0
        pb = data if getattr(data, "dataset", None) is not None else None
        x, bt = data.x.contiguous(), data.batch.contiguous()
        ei = data.edge_index.contiguous() if pb is None else None
        flags = m._flags_of(data)
        maxn = m._max_nodes_of(data)
        epoch = m._next_epoch()
        if pb is not None:
            _lib.check(L.dgcnn_assemble(pb.dataset.desc_ref, B, N, E, C, pb.ids_ptr, pb.onode_ptr, pb.oedge_ptr, ws.data_ptr(),
                                        x.data_ptr(), bt.data_ptr(), pb.y.data_ptr(), flags, maxn, epoch, stream), "dgcnn_assemble")
            flags |= _lib.FLAG_PREPARED
        _lib.check(L.dgcnn_model_forward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(),
                                         ei.data_ptr() if (E and pb is None) else None, bt.data_ptr(), ws.data_ptr(),
                                         logp.data_ptr(), training, seed, flags, maxn,
                                         int(getattr(data, "max_edges", 0) or 0), epoch, stream), "dgcnn_model_forward")
        scale = 0.0 if global_batch is None else 1.0 / float(global_batch)
        if fuse_adam:
            self.step_count += 1
            _lib.check(L.dgcnn_model_backward_step(N, E, B, F, C, flat.data_ptr(), x.data_ptr(), ws.data_ptr(),
                                                   logp.data_ptr(), y.data_ptr(), scale, training,
                                                   self.grads.data_ptr(), self.metrics.data_ptr(),
                                                   self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                   self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                                   flags, maxn, stream), "dgcnn_model_backward_step")
        else:
            _lib.check(L.dgcnn_model_backward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(), ws.data_ptr(),
                                              logp.data_ptr(), None, y.data_ptr(), scale, training,
                                              self.grads.data_ptr(), self.metrics.data_ptr(), flags, maxn, stream),
                       "dgcnn_model_backward")
        m._last_ws, m._last_dims = ws, (N, E, B, F, C)
        return logp[:B]

    def optimizer_step(self) -> None:
        """Adam + zero_grad over the flat buffer (train.py:41-42)."""
        flat = self.model.flat_params_fast()
        self.step_count += 1
        dev = flat.device
        stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
        rc = _lib.lib().dgcnn_adam_step(flat.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), flat.numel(), self.step_count, self.lr,
                                        self.betas[0], self.betas[1], self.eps, 1, stream)
        if rc != 0:
            _lib.check(rc, "dgcnn_adam_step")

    # ---- pipelined step: ONE C-ABI call per batch, next batch's graph prep on the library's side stream ----
    def _step_args(self, data, y):
        """cached ``dgcnn_step_args`` of a batch object: sizes, input pointers, layout flags and the optimizer's
        constants are filled ONCE per batch object; a step only touches epoch / seed / step / ws."""
        ent = self._args_cache.get(id(data))
        if ent is not None and ent[0] is data and ent[1] is y and self._same_tensors(ent, data):
            return ent
        m = self.model
        N, E, B, F, C = self._dims(data)
        pb = data if getattr(data, "dataset", None) is not None else None
        x, bt = data.x.contiguous(), data.batch.contiguous()
        ei = data.edge_index.contiguous() if pb is None else None
        yy = y.contiguous()
        a = _lib.StepArgs()
        a.N, a.E, a.B, a.F, a.C = N, E, B, F, C
        a.max_nodes = m._max_nodes_of(data)
        a.max_edges = int(getattr(data, "max_edges", 0) or 0)
        a.x, a.edge_index, a.batch, a.y = x.data_ptr(), (ei.data_ptr() if (E and pb is None) else None), bt.data_ptr(), yy.data_ptr()
        if pb is not None:       # batch of a PreparedDataset: the step (or the previous step's riders) assembles it, no edge list
            if yy.data_ptr() != pb.y.data_ptr():
                raise _lib.DgcnnError("a PreparedBatch is trained on its own labels (batch.y): the assembly fills that buffer")
            a.ds, a.ds_ids, a.ds_onode, a.ds_oedge = pb.dataset.desc_ref, pb.ids_ptr, pb.onode_ptr, pb.oedge_ptr
        a.lr, a.beta1, a.beta2, a.eps = self.lr, self.betas[0], self.betas[1], self.eps
        need = _lib.workspace_bytes(N, E, B, F, C)
        if len(self._args_cache) >= self.ARGS_CACHE_MAX:      # bounded: every entry keeps its batch's tensors alive
            self._args_cache.clear()
        ent = (data, y, a, need, (x, ei, bt, yy), (N, E, B, F, C), _lib.ctypes.byref(a), x.device,
               (_lib.FLAG_COALESCED_UNDIRECTED if getattr(data, "coalesced_undirected", False) else 0)
               | int(getattr(data, "mode_flags", 0) or 0),
               (data.x, data.edge_index, data.batch))
        self._args_cache[id(data)] = ent
        return ent

    @staticmethod
    def _same_tensors(ent, data) -> bool:
        """the cached pointers still describe ``data``: a caller (a PyG-style transform, say) may have REBOUND ``data.x`` /
        ``edge_index`` / ``batch`` to new tensors on the same batch object -- three identity compares per step"""
        src = ent[9]
        return data.x is src[0] and data.edge_index is src[1] and data.batch is src[2]

    def _bind_static(self, a) -> None:
        """pointers that are constant for this trainer (re-bound only if a buffer was re-allocated)"""
        a.params, a.grads, a.metrics = self._p_flat, self._p_grads, self._p_metrics

    def pipelined_step(self, data, y, next_data=None, next_y=None, global_batch: Optional[int] = None,
                       fuse_adam: bool = True, evaluate: bool = False) -> torch.Tensor:
        """``dgcnn_pipeline_train_step``: forward + backward (+ fused Adam) of ``data`` as one call; when
        ``next_data`` is given its graph structure is prepared during this step (extra workgroups on the step's
        two graph-per-workgroup launches) and the following ``pipelined_step(next_data, ...)`` skips its own
        preparation.  Bit-identical to ``train_step`` without look-ahead.  The Python side of a step is a dict
        lookup and a handful of field stores (the host must stay ahead of a ~60 us GPU step).
        ``evaluate``: ``dgcnn_pipeline_eval_step`` instead -- forward in eval mode + metrics, same look-ahead (round 5)."""
        m = self.model
        ent = self._args_cache.get(id(data))
        if ent is None or ent[0] is not data or ent[1] is not y or not self._same_tensors(ent, data):
            ent = self._step_args(data, y)
        a, need, dims, aref, dev = ent[2], ent[3], ent[5], ent[6], ent[7]
        if self._pipe is None:
            L = _lib.lib()
            h = _lib.c_void_p()
            _lib.check(L.dgcnn_pipeline_create(_lib.ctypes.byref(h)), "dgcnn_pipeline_create")
            self._pipe = h
            self._pipe_fn = L.dgcnn_pipeline_train_step
            self._pipe_eval_fn = L.dgcnn_pipeline_eval_step
        flat = m.flat_params_fast()
        if flat.data_ptr() != self._p_flat:         # model moved / re-flattened: refresh the cached pointers
            self._p_flat, self._p_grads, self._p_metrics = flat.data_ptr(), self.grads.data_ptr(), self.metrics.data_ptr()
            self._p_m, self._p_v = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        pe = self._prep_ent
        prepared = pe is ent          # the very cache entry that was prepared: same batch object AND the same tensors
        if prepared:
            slot = self._prep_slot               # its workspace was sized when it was handed in as `next`
            if a.epoch > m.__dict__.get("_epoch", 0):      # (never backwards: a step may have run between its preparation and now)
                m.__dict__["_epoch"] = a.epoch
        else:
            # (a prepared-but-abandoned batch keeps its slot untouched: use the other one)
            slot = self._cur if pe is None else 1 - self._prep_slot
            a.epoch = m._next_epoch()
        # a look-ahead that this call neither consumes nor replaces stays valid (its slot is not the one used here): an
        # eval_step between train_step(..., next_data=X) and train_step(X) does not throw X's preparation away
        if prepared or pe is None or (next_data is not None and next_data is not data):
            self._prep_ent = None
        sl = self._slots[slot]
        if sl["ws"] is None or need > sl["bytes"] or sl["ws"].device != dev:
            self._slot_ws(slot, need, dev)
        ws = sl["ws"]
        sl["dims"] = dims
        B, C = dims[2], dims[4]
        lp = self._logp
        if lp is None or lp.shape[0] < B or lp.shape[1] != C or lp.device != dev:
            lp = self._logp = torch.empty(max(B, 64), C, dtype=torch.float32, device=dev)
            self._logp_views = {}
        training = 1 if (m.training and not evaluate) else 0
        a.ws, a.logp, a.params, a.grads, a.metrics = sl["ptr"], lp.data_ptr(), self._p_flat, self._p_grads, self._p_metrics
        if self._peer is not None and not fuse_adam and not evaluate:      # one-shot route: this step's buffer of the exchange block
            a.grads = self._peer.grad_ptr(self.step_count + 1)
        a.training = training
        a.seed = m._next_seed() if training else 0
        inf = m._inference_flag() if evaluate else 0      # (forward-only use: part of the preparation's form, like the family flags)
        a.flags = ent[8] | (_lib.FLAG_PREPARED if prepared else 0) | m._mode_flags() | self._excl | inf
        a.loss_scale = 0.0 if global_batch is None else 1.0 / float(global_batch)
        if fuse_adam and not evaluate:
            self.step_count += 1
            a.step = self.step_count
            a.exp_avg, a.exp_avg_sq = self._p_m, self._p_v
        else:
            a.exp_avg = a.exp_avg_sq = None
        nref = None
        if next_data is not None and next_data is not data:
            ny = next_y if next_y is not None else next_data.y
            nent = self._args_cache.get(id(next_data))
            if nent is None or nent[0] is not next_data or nent[1] is not ny or not self._same_tensors(nent, next_data):
                nent = self._step_args(next_data, ny)
            na = nent[2]
            nsl = self._slots[1 - slot]
            if nsl["ws"] is None or nent[3] > nsl["bytes"] or nsl["ws"].device != dev:
                self._slot_ws(1 - slot, nent[3], dev)
            nsl["dims"] = nent[5]
            # (the look-ahead is prepared for the kind of step this one is; a step of the other kind prepares again itself)
            na.ws, na.flags, na.epoch = nsl["ptr"], nent[8] | m._mode_flags() | self._excl | inf, m._next_epoch()
            nref = nent[6]
            self._prep_ent, self._prep_slot = nent, 1 - slot
            self._cur = 1 - slot
        elif self._prep_ent is None:
            self._cur = slot
        rc = (self._pipe_eval_fn if evaluate else self._pipe_fn)(self._pipe, aref, nref, torch._C._cuda_getCurrentRawStream(dev.index))
        if rc != 0:
            _lib.check(rc, "dgcnn_pipeline_eval_step" if evaluate else "dgcnn_pipeline_train_step")
        self._ws = ws
        md = m.__dict__
        md["_last_ws"], md["_last_dims"] = ws, dims
        v = self._logp_views.get(B)
        if v is None:
            v = self._logp_views[B] = lp[:B]
        return v

    def train_step(self, data, y, global_batch: Optional[int] = None, next_data=None) -> torch.Tensor:
        """One iteration of the body of the reference ``train()`` loop (train.py:36-45), one C-ABI call.
        ``next_data``: the batch the NEXT call will be given (optional) -- its graph preparation then overlaps
        this step.  Data parallel: forward+backward, ONE flat-bucket RCCL all-reduce (dgcnn_amd/dist.py), Adam."""
        if self._allreduce is None:
            # single GPU: optimizer fused into the weight-gradient kernel (no separate Adam launch)
            return self.pipelined_step(data, y, next_data, None, global_batch, fuse_adam=True)
        if global_batch is None and self._dp_world > 1:
            # the loss is the mean over the GLOBAL batch (nn.NLLLoss() default, train.py:98): without the global size
            # every rank would scale by 1/B_local and the SUM all-reduce would yield world_size times the gradient.
            # One tiny all-reduce + host sync; loops with a static split pass `global_batch` and skip it.
            global_batch = self._allreduce.global_batch(_batch_size_of(data), data.x.device)
        if self._one_shot:
            if self._peer is None:
                from .dist import PeerExchange
                self._peer = PeerExchange(self.model.flat_params.numel(), self.pg, data.x.device)
            logp = self.pipelined_step(data, y, next_data, None, global_batch, fuse_adam=False)
            self.step_count += 1
            flat = self.model.flat_params_fast()
            dev = flat.device
            self._peer.step(self.step_count, flat, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas,
                            self.eps, torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))
            return logp
        logp = self.pipelined_step(data, y, next_data, None, global_batch, fuse_adam=False)
        self._allreduce(self.grads)
        self.optimizer_step()
        return logp

    @torch.no_grad()
    def eval_step(self, data, y, global_batch: Optional[int] = None, next_data=None) -> torch.Tensor:
        """Body of the reference ``test()`` loop (train.py:59-64) as ONE C call (``dgcnn_pipeline_eval_step``): forward in
        eval mode + loss / #correct folded into the device-side metrics accumulator -- one launch where the batch admits the
        one-launch evaluation kernel (``DGCNN_FORM_EVAL``).  ``next_data``: the batch the NEXT call (``eval_step`` or
        ``train_step``) will be given -- its graph preparation then overlaps this one, as in training.  A look-ahead handed to an
        EARLIER call and not yet consumed (``train_step(A, next_data=X)``, ``eval_step(Y)``, ``train_step(X)``) stays prepared
        unless this call is given a ``next_data`` of its own, which replaces it.  Data parallel: the
        batch loss is scaled by 1/``global_batch`` (derived with one small all-reduce when not given) so that the ranks'
        contributions add up to the global batch mean."""
        if global_batch is None and self._dp_world > 1 and self._allreduce is not None:
            global_batch = self._allreduce.global_batch(_batch_size_of(data), data.x.device)
        return self.pipelined_step(data, y, next_data, None, global_batch, fuse_adam=False, evaluate=True)

    def reset_metrics(self) -> None:
        self.metrics.zero_()

    def read_metrics(self) -> Tuple[float, float]:
        """(sum of per-batch mean losses, number correct) -- ONE host sync.  Under a process group the two numbers are
        summed over the ranks (every rank scaled its losses by 1/B_global), i.e. the dataset-level values the
        reference returns (/root/reference/train.py:47,66), identical on every rank.  Also surfaces every input
        error the kernels flagged since the previous call, in any batch (the error words are epoch-tagged and
        sticky per workspace slot), and a non-finite loss (a label outside [0, C) poisons the accumulator with
        NaN -- the reference's NLLLoss raises for it)."""
        m = self.metrics
        if self.pg is not None and self._dp_world > 1:
            import torch.distributed as dist
            local = m.clone()
            m = local.clone()
            dist.all_reduce(m, op=dist.ReduceOp.SUM, group=self.pg)
            lv = float(local[0])
            if not math.isfinite(lv):      # name the shard: after the reduction every rank sees the same poisoned sum
                raise _lib.DgcnnError(f"non-finite loss on rank {dist.get_rank(self.pg)}: a label outside [0, num_classes) "
                                      "or diverged parameters")
        # ONE device-to-host copy: the two metrics (as their bit patterns) and the four error words of every workspace slot in
        # use are concatenated on the device (three separate .tolist() / .cpu() calls were three syncs of ~100 us each per epoch:
        # 15 us per batch of a 20-batch epoch)
        slots = [sl for sl in self._slots if sl.get("dims") and sl["ws"] is not None]
        if slots and self.model._last_dims is not None and m.is_cuda:
            words = torch.cat([m.view(torch.int32)] + [_lib.ws_view(sl["ws"], "err", *sl["dims"]) for sl in slots]).tolist()
            v = torch.tensor(words[:2], dtype=torch.int32).view(torch.float32).tolist()
            for k in range(len(slots)):
                self.model._check_err_words(words[2 + 8 * k: 10 + 8 * k], self._err_checked, self.model._epoch)
            self._err_checked = self.model._epoch
        else:
            v = m.tolist()
        if self._peer is not None:
            self._peer.check()
        if not math.isfinite(v[0]):          # NaN (poisoned label) and +-inf (divergence) alike
            raise _lib.DgcnnError("non-finite loss: a label outside [0, num_classes) or diverged parameters")
        return float(v[0]), float(v[1])

    # ---- epoch loops with the reference's return values ------------------------------------
    def train_epoch(self, batches: Iterable, num_samples: int, global_batch=None) -> Tuple[float, float]:
        """``train()`` of train.py:27-47: returns (running_loss/num_batches, correct/num_samples*100).
        Data parallel: ``batches`` are this rank's shards, ``num_samples`` the GLOBAL sample count, ``global_batch``
        the global size of every batch (int, or a sequence with one entry per batch; None = derived per batch with a
        small all-reduce); the returned numbers are the global ones on every rank."""
        self.model.train()
        self.reset_metrics()
        nb = 0
        it = iter(batches)
        cur = next(it, None)
        while cur is not None:          # one batch of look-ahead: batch i+1's graph prep overlaps step i
            nxt = next(it, None)
            gb = global_batch if global_batch is None or isinstance(global_batch, int) else global_batch[nb]
            self.train_step(cur, cur.y, global_batch=gb, next_data=nxt)
            nb += 1
            cur = nxt
        loss, correct = self.read_metrics()
        return loss / max(nb, 1), correct / max(num_samples, 1) * 100.0

    def test_epoch(self, batches: Iterable, num_samples: int) -> Tuple[float, float]:
        """``test()`` of train.py:49-66."""
        self.reset_metrics()
        nb = 0
        it = iter(batches)
        cur = next(it, None)
        while cur is not None:          # one batch of look-ahead, as in train_epoch
            nxt = next(it, None)
            self.eval_step(cur, cur.y, next_data=nxt)
            nb += 1
            cur = nxt
        loss, correct = self.read_metrics()
        return loss / max(nb, 1), correct / max(num_samples, 1) * 100.0
