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
build's ``Model`` + torch's Adam) also works and is what tests/test_gpu_dropin.py exercises.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch

from . import _lib
from .model import Model, _batch_size_of


class Trainer:
    """Fused train/eval step for a :class:`dgcnn_amd.Model`.

    lr/betas/eps default to ``torch.optim.Adam`` defaults, which is what the reference uses
    (``Adam(model.parameters())``, /root/reference/train.py:99).
    ``process_group``: when given (data parallel, one process per GPU), gradients are summed over
    ranks with ONE all-reduce of the flat buffer per step and the loss is scaled by the GLOBAL
    batch size, so N ranks x B/N graphs reproduce one rank x B graphs.
    """

    def __init__(self, model: Model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None):
        self.model = model
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.pg = process_group
        from .dist import GradAllReduce
        self._allreduce = GradAllReduce(process_group) if process_group is not None else None
        if self._allreduce is not None and self._allreduce.world_size == 1 and not __import__('os').environ.get('BENCH_FORCE_DIST'):
            self._allreduce = None           # a 1-rank group needs no collective
        self.step_count = 0
        flat = model.flat_params
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.grads = torch.zeros_like(flat)
        self.metrics = torch.zeros(2, dtype=torch.float32, device=flat.device)
        self._logp = None
        # two workspace slots: the step runs in slot `_cur`; `prefetch` prepares the next batch's graph structure
        # in the other slot on a side stream (software pipelining of graph prep across steps)
        self._slots = [{"ws": None, "bytes": 0, "free": None}, {"ws": None, "bytes": 0, "free": None}]
        self._cur = 0
        self._side = None
        self._ready_evt = None
        self._pf = None          # (data, N, E, B, F, C, flags, epoch, ...) of the batch prepared in slot 1-_cur

    # ---- buffers reused across steps (sizes only grow) -------------------------------------
    def _slot_ws(self, k, need, device):
        sl = self._slots[k]
        if sl["ws"] is None or need > sl["bytes"] or sl["ws"].device != device:
            sl["ws"] = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=device)
            sl["bytes"] = sl["ws"].numel()
        return sl["ws"]

    def _buffers(self, N, E, B, F, C, device):
        need = _lib.workspace_bytes(N, E, B, F, C)
        self._ws = self._slot_ws(self._cur, need, device)
        if self._logp is None or self._logp.shape[0] < B or self._logp.shape[1] != C or self._logp.device != device:
            self._logp = torch.empty(max(B, 64), C, dtype=torch.float32, device=device)
        return self._ws, self._logp

    def prefetch(self, data) -> None:
        """Prepare ``data``'s graph structure (CSR, degrees, graph ranges) NOW, on a side stream, into the spare
        workspace, so that the next ``train_step(data)`` / ``forward_backward(data)`` skips graph prep.
        Graph prep depends on the batch only (not on the parameters), so a training loop calls this for batch
        i+1 right after launching step i -- what a DataLoader with prefetching does on the host, done here on
        the device.  Purely an overlap: every step's prep is still executed, once."""
        L = _lib.lib()
        m = self.model
        N, E, B, F, C = self._dims(data)
        dev = data.x.device
        need = _lib.workspace_bytes(N, E, B, F, C)
        cur = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
            self._ready_evt = torch.cuda.Event()
        o = 1 - self._cur
        ws = self._slot_ws(o, need, dev)
        flags = m._flags_of(data)
        epoch = m._next_epoch()
        sl = self._slots[o]
        if sl["free"] is not None:
            self._side.wait_event(sl["free"])            # that slot's previous consumer (step i-1) is done
        else:
            self._side.wait_stream(cur)                  # first use: order after whatever allocated/queued so far
        ei, bt = data.edge_index.contiguous(), data.batch.contiguous()
        _lib.check(L.dgcnn_model_prepare(N, E, B, F, C, ei.data_ptr() if E else None, bt.data_ptr(),
                                         ws.data_ptr(), flags, epoch, self._side.cuda_stream),
                   "dgcnn_model_prepare")
        self._ready_evt.record(self._side)
        self._pf = (data, N, E, B, F, C, flags, epoch, ei, bt)

    def _dims(self, data):
        x, ei = data.x, data.edge_index
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
        x, ei, bt = data.x.contiguous(), data.edge_index.contiguous(), data.batch.contiguous()
        flags = m._flags_of(data)
        if self._pf is not None and self._pf[0] is data and self._pf[1:6] == (N, E, B, F, C):
            # consume the prefetched graph structure: swap workspaces, wait for the side stream, skip prep
            _, _, _, _, _, _, pflags, epoch, _, _ = self._pf
            self._pf = None
            self._cur = 1 - self._cur
            ws = self._ws = self._slots[self._cur]["ws"]
            cur.wait_event(self._ready_evt)
            flags = pflags | _lib.FLAG_PREPARED
            m._epoch = epoch
            swapped = True
        else:
            epoch = m._next_epoch()
            swapped = False
        _lib.check(L.dgcnn_model_forward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(),
                                         ei.data_ptr() if E else None, bt.data_ptr(), ws.data_ptr(),
                                         logp.data_ptr(), training, seed, flags, m._max_nodes_of(data),
                                         int(getattr(data, "max_edges", 0) or 0), epoch, stream), "dgcnn_model_forward")
        scale = 0.0 if global_batch is None else 1.0 / float(global_batch)
        if fuse_adam:
            self.step_count += 1
            _lib.check(L.dgcnn_model_backward_step(N, E, B, F, C, flat.data_ptr(), x.data_ptr(), ws.data_ptr(),
                                                   logp.data_ptr(), y.data_ptr(), scale, training,
                                                   self.grads.data_ptr(), self.metrics.data_ptr(),
                                                   self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                   self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                                   stream), "dgcnn_model_backward_step")
        else:
            _lib.check(L.dgcnn_model_backward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(), ws.data_ptr(),
                                              logp.data_ptr(), None, y.data_ptr(), scale, training,
                                              self.grads.data_ptr(), self.metrics.data_ptr(), stream),
                       "dgcnn_model_backward")
        m._last_ws, m._last_dims = ws, (N, E, B, F, C)
        if self._side is not None:      # prefetching in use: mark when this step's slot becomes reusable
            sl = self._slots[self._cur]
            if sl["free"] is None:
                sl["free"] = torch.cuda.Event()
            sl["free"].record(cur)
        return logp[:B]

    def optimizer_step(self) -> None:
        """Adam + zero_grad over the flat buffer (train.py:41-42)."""
        L = _lib.lib()
        flat = self.model.flat_params
        self.step_count += 1
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        _lib.check(L.dgcnn_adam_step(flat.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                     self.exp_avg_sq.data_ptr(), flat.numel(), self.step_count, self.lr,
                                     self.betas[0], self.betas[1], self.eps, 1, stream), "dgcnn_adam_step")

    def train_step(self, data, y, global_batch: Optional[int] = None) -> torch.Tensor:
        """One iteration of the body of the reference ``train()`` loop (train.py:36-45)."""
        if self._allreduce is None:
            # single GPU: optimizer fused into the weight-gradient kernel (no separate Adam launch)
            return self.forward_backward(data, y, global_batch, fuse_adam=True)
        logp = self.forward_backward(data, y, global_batch)
        self._allreduce(self.grads)              # ONE flat-bucket RCCL all-reduce per step (dgcnn_amd/dist.py)
        self.optimizer_step()
        return logp

    @torch.no_grad()
    def eval_step(self, data, y) -> torch.Tensor:
        """Body of the reference ``test()`` loop (train.py:59-64): forward only + metrics."""
        L = _lib.lib()
        m = self.model
        was = m.training
        m.eval()
        try:
            N, E, B, F, C = self._dims(data)
            dev = data.x.device
            ws, logp = self._buffers(N, E, B, F, C, dev)
            flat = m.flat_params
            stream = torch.cuda.current_stream(dev).cuda_stream
            x, ei, bt = data.x.contiguous(), data.edge_index.contiguous(), data.batch.contiguous()
            _lib.check(L.dgcnn_model_forward(N, E, B, F, C, flat.data_ptr(), x.data_ptr(),
                                             ei.data_ptr() if E else None, bt.data_ptr(), ws.data_ptr(),
                                             logp.data_ptr(), 0, 0, m._flags_of(data), m._max_nodes_of(data),
                                             int(getattr(data, "max_edges", 0) or 0), m._next_epoch(), stream),
                       "dgcnn_model_forward")
            m._last_ws, m._last_dims = ws, (N, E, B, F, C)
            lp = logp[:B]
            # metrics with plain torch ops (evaluation is not the timed hot path)
            self.metrics[0] += -lp.gather(1, y.view(-1, 1)).mean()
            self.metrics[1] += (lp.argmax(dim=1) == y).sum()
        finally:
            m.train(was)
        return lp

    def reset_metrics(self) -> None:
        self.metrics.zero_()

    def read_metrics(self) -> Tuple[float, float]:
        """(sum of per-batch mean losses, number correct) -- ONE host sync; also surfaces any input
        error the kernels flagged for the most recent batch."""
        v = self.metrics.tolist()
        self.model.check_errors()
        return float(v[0]), float(v[1])

    # ---- epoch loops with the reference's return values ------------------------------------
    def train_epoch(self, batches: Iterable, num_samples: int) -> Tuple[float, float]:
        """``train()`` of train.py:27-47: returns (running_loss/num_batches, correct/num_samples*100)."""
        self.model.train()
        self.reset_metrics()
        nb = 0
        for b in batches:
            self.train_step(b, b.y)
            nb += 1
        loss, correct = self.read_metrics()
        return loss / max(nb, 1), correct / max(num_samples, 1) * 100.0

    def test_epoch(self, batches: Iterable, num_samples: int) -> Tuple[float, float]:
        """``test()`` of train.py:49-66."""
        self.reset_metrics()
        nb = 0
        for b in batches:
            self.eval_step(b, b.y)
            nb += 1
        loss, correct = self.read_metrics()
        return loss / max(nb, 1), correct / max(num_samples, 1) * 100.0
