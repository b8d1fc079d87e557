"""``Adam`` with the call signature of ``torch.optim.Adam`` that updates a :class:`dgcnn_amd.Model` with ONE kernel.

The reference builds its optimizer as ``Adam(model.parameters())`` (/root/reference/train.py:11,99).  With this
build's ``Model`` all 16 parameters are views of one flat buffer and ``loss.backward()`` leaves their ``.grad`` as views
of one flat gradient buffer in the same layout, so ``optimizer.step()`` can be a single ``dgcnn_adam_step`` launch over
the flat buffers instead of torch's multi-tensor path (~10 launches and ~190 us of host time per step for 16 small
tensors).  Changing the import is the only edit to the reference loop::

    from dgcnn_amd.optim import Adam          # instead of: from torch.optim import Adam

Semantics are those of ``torch.optim.Adam`` defaults (no weight decay, no amsgrad; `step`, `exp_avg`, `exp_avg_sq`
per parameter in ``state_dict()``).  Whenever the flat layout is NOT recognised (other modules' parameters, gradients
that are not views of one buffer, a parameter without gradient, ...) the step falls back to torch's own implementation
for that call -- same numbers, just slower.  GPU tensors only take the fused route.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import _lib


def _global_step_hooks() -> bool:
    from torch.optim import optimizer as _o
    return bool(getattr(_o, "_global_optimizer_pre_hooks", None)) or bool(getattr(_o, "_global_optimizer_post_hooks", None))


class Adam(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False):
        if weight_decay != 0.0 or amsgrad:
            raise ValueError("dgcnn_amd.optim.Adam implements torch.optim.Adam's defaults (weight_decay=0, amsgrad=False); "
                             "use torch.optim.Adam for the other variants")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._flat = {}          # id(group) -> [params_flat, exp_avg, exp_avg_sq, offsets, step tensor, step] once recognised

    # ---- layout recognition ---------------------------------------------------------------------------
    @staticmethod
    def _flat_of(tensors):
        """the 1-D tensor spanning the storage range [min offset, max end) that all `tensors` are views of, plus
        each tensor's offset inside it -- or None when they do not share one storage / overlap / are not dense"""
        t0 = tensors[0]
        st = t0.untyped_storage()
        base, spans = st.data_ptr(), []
        for t in tensors:
            if t.untyped_storage().data_ptr() != base or not t.is_contiguous() or t.dtype != torch.float32 \
                    or t.device != t0.device:
                return None
            spans.append((t.storage_offset(), t.numel()))
        lo = min(o for o, _ in spans)
        hi = max(o + n for o, n in spans)
        order = sorted(spans)
        for (o1, n1), (o2, _) in zip(order, order[1:]):
            if o1 + n1 > o2:
                return None
        if (hi - lo) > 2 * sum(n for _, n in spans) + 64:        # not one packed buffer
            return None
        flat = torch.empty(0, dtype=torch.float32, device=t0.device).set_(st, lo, (hi - lo,))
        return flat, [o - lo for o, _ in spans]

    def _state_views(self, group, pflat, offs):
        m = torch.zeros_like(pflat)
        v = torch.zeros_like(pflat)
        for p, off in zip(group["params"], offs):
            stt = self.state[p]
            prev_m, prev_v = stt.get("exp_avg"), stt.get("exp_avg_sq")
            mv = m[off:off + p.numel()].view(p.shape)
            vv = v[off:off + p.numel()].view(p.shape)
            if prev_m is not None:                     # continue from a loaded / torch-produced state
                mv.copy_(prev_m); vv.copy_(prev_v)
            stt["exp_avg"], stt["exp_avg_sq"] = mv, vv
        # ONE step counter object shared by the group's parameters (torch keeps one per parameter; a state_dict saved
        # from here holds 16 equal values and loads back either way)
        prev = self.state[group["params"][0]].get("step")
        shared = torch.tensor(float(prev) if prev is not None else 0.0, dtype=torch.float32)
        for p in group["params"]:
            self.state[p]["step"] = shared
        return m, v, shared

    def load_state_dict(self, state_dict) -> None:
        """As ``Optimizer.load_state_dict``; the cached flat moment buffers are dropped so that the next ``step()``
        rebuilds them FROM the loaded ``exp_avg`` / ``exp_avg_sq`` / ``step`` (a resumed run must not continue from
        the moments it had before the load)."""
        super().load_state_dict(state_dict)
        self._flat.clear()

    def zero_grad(self, set_to_none: bool = True) -> None:
        """``Optimizer.zero_grad``; the default (``set_to_none=True``) without torch's profiler range and per-device grouping"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    # ---- step -----------------------------------------------------------------------------------------
    def step(self, closure=None):
        # (torch wraps every optimizer's ``step`` in a profiler range + hook dispatch, ~25 us of host time per call: this
        #  method is marked ``hooked`` so that the wrapper is not installed, and takes the wrapped route itself whenever a
        #  step hook is registered, on the optimizer or globally)
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _global_step_hooks():
            return self._hooked_step(closure)
        return self._step_impl(closure)

    step.hooked = True

    def _hooked_step(self, closure=None):
        fn = self.__dict__.get("_wrapped_impl")
        if fn is None:
            fn = self.__dict__["_wrapped_impl"] = Optimizer.profile_hook_step(Adam._step_impl)
        return fn(self, closure)

    def _step_impl(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params = group["params"]
            if not self._fused_step(group, params):
                with torch.no_grad():
                    self._torch_step(group, params)
        return loss

    def _fused_step(self, group, params) -> bool:
        if not params:
            return False
        key = id(group)
        ent = self._flat.get(key)
        p0 = params[0]
        if ent is None or ent[0].data_ptr() != p0.data_ptr() - 4 * ent[3][0] or ent[6] != len(params):
            if not p0.is_cuda or any(p.grad is None for p in params):
                return False
            rec = self._flat_of(params)
            if rec is None:
                return False
            pflat, offs = rec
            with torch.no_grad():
                m, v, shared = self._state_views(group, pflat, offs)
            ent = self._flat[key] = [pflat, m, v, offs, shared, int(shared.item()), len(params)]
        pflat, m, v, offs, shared, nstep, _ = ent
        if self.state[p0]["step"] is not shared:       # a state_dict was loaded: adopt its counter
            shared.fill_(float(self.state[p0]["step"]))
            nstep = ent[5] = int(shared.item())
            for p in params:
                self.state[p]["step"] = shared
        # parameters AND gradients must mirror the flat layout: parameter i at pflat + offs[i] (a parameter re-pointed by hand
        # since the layout was recognised fails here), its gradient at the same offset of ONE gradient buffer, dense fp32
        g0 = p0.grad
        if g0 is None:
            return False
        pbase = pflat.data_ptr()
        gbase = g0.data_ptr() - 4 * offs[0]
        f32 = torch.float32
        for p, off in zip(params, offs):
            g = p.grad
            if g is None or g.data_ptr() != gbase + 4 * off or p.data_ptr() != pbase + 4 * off or g.dtype is not f32 \
                    or not g.is_contiguous():
                return False
        # ... and that buffer is one allocation spanning the whole layout (the alignment gaps included)
        n = pflat.numel()
        gst = g0.untyped_storage()
        lo = gbase - gst.data_ptr()
        if lo < 0 or lo + 4 * n > gst.nbytes() or g0.device != pflat.device:
            return False
        step = nstep + 1
        b1, b2 = group["betas"]
        stream = torch.cuda.current_stream(pflat.device).cuda_stream
        # the gaps between segments hold zeros in both buffers (zero gradient -> zero update), so one launch over the
        # whole span is exact
        _lib.check(_lib.lib().dgcnn_adam_step(pbase, gbase, m.data_ptr(), v.data_ptr(), n, step,
                                              float(group["lr"]), float(b1), float(b2), float(group["eps"]), 0, stream),
                   "dgcnn_adam_step")
        shared += 1
        ent[5] = step
        return True

    def _torch_step(self, group, params) -> None:
        """torch's own multi-tensor Adam on this group (layout not recognised)"""
        from torch.optim.adam import adam as _adam
        # a group that HAS taken the flat route shares one step counter between its parameters: torch's multi-tensor Adam would
        # increment that one tensor once per parameter.  Back to one counter per parameter (the moments stay views of the flat
        # buffers, which is fine), and the flat entry is dropped: the next step whose layout checks out rebuilds it from the state.
        ent = self._flat.pop(id(group), None)
        if ent is not None:
            for p in params:
                self.state[p]["step"] = ent[4].clone()
        with_grad = [p for p in params if p.grad is not None]
        if not with_grad:
            return
        grads, ms, vs, steps = [], [], [], []
        for p in with_grad:
            stt = self.state[p]
            if "exp_avg" not in stt:
                stt["step"] = torch.tensor(0.0, dtype=torch.float32)
                stt["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                stt["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            grads.append(p.grad); ms.append(stt["exp_avg"]); vs.append(stt["exp_avg_sq"]); steps.append(stt["step"])
        b1, b2 = group["betas"]
        _adam(with_grad, grads, ms, vs, [], steps, amsgrad=False, beta1=b1, beta2=b2, lr=group["lr"], weight_decay=0.0,
              eps=group["eps"], maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
              grad_scale=None, found_inf=None, has_complex=False)
