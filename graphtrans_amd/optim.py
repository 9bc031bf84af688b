"""FusedAdamW: torch.optim.AdamW semantics (main.py:178 of the reference builds
optim.AdamW(model.parameters(), lr, weight_decay)) with the whole step as ONE HIP launch per
parameter group (csrc/optim.hip).  Same constructor arguments, param_groups (LR schedulers work),
state_dict layout ({'step', 'exp_avg', 'exp_avg_sq'} per parameter).  fp32 GPU parameters only."""
import ctypes as C

import torch

from . import _lib
from .graph import _stream


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, max_grad_norm=None):
        """max_grad_norm: clip the global gradient norm like `torch.nn.utils.clip_grad_norm_(model.parameters(),
        args.grad_clip)` before the step (trainers/base_trainer.py:34-35) — two small launches, the coefficient is
        applied inside the AdamW kernel and `.grad` itself is left unscaled; `last_grad_norm` holds the norm."""
        if amsgrad:
            raise NotImplementedError("FusedAdamW: amsgrad is not supported")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay:
            raise ValueError("lr, eps and weight_decay must be non-negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid betas {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._plans = {}
        if max_grad_norm is not None and not max_grad_norm > 0:
            raise ValueError("max_grad_norm must be positive")
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm = None
        self._clip_buf = None

    # ---- per-group launch plan (device tables), rebuilt when a parameter's storage moves ----------
    def _plan(self, gi, group):
        params = [p for p in group["params"] if p.requires_grad]
        ptrs = tuple(p.data_ptr() for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan["ptrs"] == ptrs:
            return plan
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedAdamW: contiguous fp32 GPU parameters only (no CPU fallback)")
        dev = params[0].device
        chunk = _lib.lib().gt_adamw_chunk_elems()
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # 16-byte aligned segments
        if plan is not None:  # storage moved: carry the host-side step counts over
            for p, n in zip(plan["params"], plan["steps"]):
                self.state[p]["step"] = torch.tensor(float(n), dtype=torch.float32)
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        table, ctens, cloc, tensor_chunk0, steps = [], [], [], [], []
        for t, (p, off) in enumerate(zip(params, offs)):
            n = p.numel()
            st = self.state[p]
            mv, vv = m[off:off + n].view_as(p), v[off:off + n].view_as(p)
            if "exp_avg" in st:  # keep loaded / previous moments
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = mv, vv
            steps.append(int(float(st["step"])) if "step" in st else 0)
            st.setdefault("step", torch.tensor(0.0, dtype=torch.float32))
            table.append([p.data_ptr(), mv.data_ptr(), vv.data_ptr(), n])
            tensor_chunk0.append(len(ctens))
            nchunks = (n + chunk - 1) // chunk
            ctens.extend([t] * nchunks)
            cloc.extend(range(nchunks))
        tensor_chunk0.append(len(ctens))
        plan = dict(ptrs=ptrs, params=params, m=m, v=v,
                    table=torch.tensor(table, dtype=torch.int64).to(dev),
                    chunk_tensor=torch.tensor(ctens, dtype=torch.int32).to(dev),
                    chunk_local=torch.tensor(cloc, dtype=torch.int32).to(dev),
                    tensor_chunk0=tensor_chunk0, steps=steps)
        self._plans[gi] = plan
        return plan

    def step(self, closure=None):
        """torch wraps Optimizer.step in a profiler record_function whose exit alone costs ~0.7 ms of host
        time per step here; this step is marked `hooked` so it is left alone, and runs the registered
        step pre/post hooks itself."""
        for hook in self._optimizer_step_pre_hooks.values():
            hook(self, (self,), {"closure": closure} if closure is not None else {})
        with torch.no_grad():
            loss = self._step(closure)
        for hook in self._optimizer_step_post_hooks.values():
            hook(self, (self,), {"closure": closure} if closure is not None else {})
        return loss

    step.hooked = True

    def _step(self, closure=None):
        from . import w3
        w3.weights_changed()   # the kernel writes the parameters through raw pointers: cached bf16x3 weight images are stale
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        MAXT = 384  # GT_ADAMW_MAX_TENSORS
        work = []
        for gi, group in enumerate(self.param_groups):
            if not any(p.requires_grad for p in group["params"]):
                continue
            plan = self._plan(gi, group)
            params = plan["params"]
            grads = [p.grad for p in params]
            fast = plan.get("fast")
            # the SAME gradient tensors as last step (the fused model path hands out persistent views of its flat buffer): they were
            # validated then -- 4 attribute reads per tensor less, ~0.1 ms of host time per step on a 128-tensor model
            same = fast is not None and len(fast["refs"]) == len(grads) and all(r() is g for r, g in zip(fast["refs"], grads))
            plan["same"] = same
            if not same:
                for t, g in enumerate(grads):
                    if g is None:
                        continue
                    if g.dtype != torch.float32 or not g.is_cuda or g.is_sparse:
                        raise RuntimeError("FusedAdamW: dense fp32 GPU gradients only")
                    if not g.is_contiguous():
                        grads[t] = g.contiguous()
            work.append((group, plan, grads))
        scale_ptr = None
        if self.max_grad_norm is not None and work:
            total = sum(plan["tensor_chunk0"][-1] for _, plan, _ in work)
            dev = work[0][1]["m"].device
            if self._clip_buf is None or self._clip_buf.numel() < total + 2 or self._clip_buf.device != dev:
                self._clip_buf = torch.empty(total + 2, dtype=torch.float32, device=dev)
            buf, off = self._clip_buf, 0
            for _, plan, grads in work:
                n = len(plan["params"])
                for t0 in range(0, n, MAXT):
                    t1 = min(t0 + MAXT, n)
                    arr = (C.c_void_p * (t1 - t0))(*[(g.data_ptr() if g is not None else None) for g in grads[t0:t1]])
                    c0, c1 = plan["tensor_chunk0"][t0], plan["tensor_chunk0"][t1]
                    _lib.launch("gt_grad_sqnorm", plan["table"].data_ptr(), plan["chunk_tensor"].data_ptr(),
                                plan["chunk_local"].data_ptr(), c0, c1 - c0, t0, t1 - t0, arr, buf.data_ptr() + 4 * off, _stream())
                off += plan["tensor_chunk0"][-1]
            _lib.launch("gt_grad_clip_coef", buf.data_ptr(), total, float(self.max_grad_norm), buf.data_ptr() + 4 * total, _stream())
            self.last_grad_norm = buf[total]
            scale_ptr = buf.data_ptr() + 4 * (total + 1)
        for group, plan, grads in work:
            params, steps = plan["params"], plan["steps"]
            beta1, beta2 = group["betas"]
            # fast path (every step of a normal run): every parameter has a gradient, all share one step count, and the
            # gradient tensors are the SAME objects as last step (the fused model path hands out persistent views of its flat
            # buffer) -> the ctypes pointer arrays of the last step are reused; building them cost 0.2 ms of host time per step
            fast = plan.get("fast")
            if plan.get("same") and fast["uniform"]:
                step = steps[0] + 1
                for t in range(len(steps)):
                    steps[t] = step
                for (t0, t1, c0, c1), arr in zip(fast["chunks"], fast["arrs"]):
                    _lib.launch("gt_adamw_step", plan["table"].data_ptr(), plan["chunk_tensor"].data_ptr(),
                                plan["chunk_local"].data_ptr(), c0, c1 - c0, t0, t1 - t0, arr, float(group["lr"]),
                                float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), step,
                                scale_ptr, _stream())
                continue
            # parameters normally share one step count; those that skipped steps (no gradient, as torch
            # skips them) form extra launches
            by_step = {}
            for t, g in enumerate(grads):
                if g is None:
                    continue
                steps[t] += 1
                by_step.setdefault(steps[t], []).append(t)
            uniform = len(by_step) == 1 and all(g is not None for g in grads)
            keep_chunks, keep_arrs = [], []
            for step, tensors in by_step.items():
                live = set(tensors) if len(by_step) > 1 else None
                for t0 in range(0, len(params), MAXT):
                    t1 = min(t0 + MAXT, len(params))
                    arr = (C.c_void_p * (t1 - t0))(*[
                        (grads[t].data_ptr() if grads[t] is not None and (live is None or t in live) else None)
                        for t in range(t0, t1)])
                    c0, c1 = plan["tensor_chunk0"][t0], plan["tensor_chunk0"][t1]
                    keep_chunks.append((t0, t1, c0, c1))
                    keep_arrs.append(arr)
                    _lib.launch("gt_adamw_step", plan["table"].data_ptr(), plan["chunk_tensor"].data_ptr(),
                                plan["chunk_local"].data_ptr(), c0, c1 - c0, t0, t1 - t0, arr, float(group["lr"]),
                                float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), step,
                                scale_ptr, _stream())
            import weakref
            try:   # weak references: the cache must not keep last step's gradients alive
                plan["fast"] = dict(refs=[weakref.ref(g) for g in grads], uniform=uniform, chunks=keep_chunks, arrs=keep_arrs) if uniform else None
            except TypeError:
                plan["fast"] = None
        return loss

    def _sync_step_tensors(self):
        for plan in self._plans.values():
            for p, n in zip(plan["params"], plan["steps"]):
                self.state[p]["step"] = torch.tensor(float(n), dtype=torch.float32)

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}  # moments are re-packed into flat buffers (and step counts re-read) on the next step
        for st in self.state.values():  # torch aliases same-device tensors of the loaded dict: take copies
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].clone()
