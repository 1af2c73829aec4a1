"""GroupedAdam: torch.optim.Adam whose step() is ONE HIP launch for all parameter groups (csplat_adam_step).

The reference builds one Adam parameter group per Gaussian attribute with its own learning rate
(/root/reference/scene_reconstruction/gaussian_mesh.py:126-136, gaussian_model.py:150-160) and steps it every iteration
(train_utils.py:310-319).  torch's foreach implementation is applied group by group: 7 groups x ~8 elementwise launches for
~25 us of memory traffic.  This subclass keeps torch's state layout (`state[p] = {step, exp_avg, exp_avg_sq}`), so
`state_dict()` / `load_state_dict()`, learning-rate schedules that edit `param_groups[i]['lr']` and the reference's
Adam-state surgery during densification / pruning keep working unchanged; anything the kernel does not cover (weight decay,
amsgrad, maximize, CPU / non-fp32 / non-contiguous tensors) falls back to torch's own step for the whole call."""
import ctypes as C

import torch

from . import native as _n


class GroupedAdam(torch.optim.Adam):
    def _fusable(self):
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False) or \
                    group.get("capturable", False) or group.get("differentiable", False):
                return False
            if isinstance(group["lr"], torch.Tensor):
                return False
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32 and
                        not g.is_sparse and g.device == p.device):
                    _n.composed_fallback("GroupedAdam.step", "dtype" if (p.dtype != torch.float32 or g.dtype != torch.float32) else "layout", p)
                    return False
        return True

    def state_dict(self):
        """torch's state dict, with every moment that is a VIEW into a larger buffer (the capacity buffers of csplat/store.py after
        the first densification) replaced by a compact copy: `torch.save` serialises a tensor's whole storage, i.e. the spare
        capacity and its stale rows would otherwise travel in every checkpoint (gaussian_model.py:64-75 saves this dict)."""
        sd = super().state_dict()
        for st in sd["state"].values():
            for k, v in st.items():
                if torch.is_tensor(v) and v.numel() and v.untyped_storage().nbytes() > v.numel() * v.element_size():
                    st[k] = v.clone()
        return sd

    def _step_count(self, st):
        """the parameter's step count AFTER this step, as a Python int.  torch keeps it as a CPU tensor in the state (that is what
        state_dict() saves), and reading it back with .item() costs more than the rest of this function: a mirror keyed on the tensor
        OBJECT and its in-place version supplies it (a loaded state dict brings new tensors, an in-place edit bumps the version: both
        are read back once)."""
        t = st["step"]
        mirror = self.__dict__.setdefault("_py_steps", {})
        if len(mirror) > 4096:      # (entries of replaced state tensors are never looked up again)
            mirror.clear()
        hit = mirror.get(id(t))
        # (valid while it is the same tensor object and nobody but our own _foreach_add_ has written to it since)
        n = (hit[1] if hit is not None and hit[0] is t and hit[2] == t._version else int(t.item())) + 1
        mirror[id(t)] = (t, n, t._version + 1)
        return n

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not self._fusable():
            self.__dict__.pop("_py_steps", None)
            return super().step(closure)
        buckets = {}
        keep = []
        steps = []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            bkey = (float(beta1), float(beta2), float(group["eps"]))
            lr = float(group["lr"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                n = self._step_count(st)
                steps.append(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32):
                    raise RuntimeError("GroupedAdam: optimizer state must be contiguous fp32")
                keep.append(g)
                buckets.setdefault((p.device,) + bkey + (n,), []).append((p, g, m, v, lr))
        if steps:
            torch._foreach_add_(steps, 1)              # the state's own counters, all in one call
        for (dev, beta1, beta2, eps, step), items in buckets.items():
            n = len(items)
            arr = lambda k: (C.c_void_p * n)(*[it[k].data_ptr() for it in items])  # noqa: E731
            numel = (C.c_int64 * n)(*[it[0].numel() for it in items])
            lrs = (C.c_double * n)(*[it[4] for it in items])
            with torch.cuda.device(dev):
                _n.check(_n.lib.csplat_adam_step(_n.stream_handle(dev), n, C.cast(arr(0), C.c_void_p), C.cast(arr(1), C.c_void_p),
                                                 C.cast(arr(2), C.c_void_p), C.cast(arr(3), C.c_void_p), C.cast(numel, C.c_void_p),
                                                 C.cast(lrs, C.c_void_p), beta1, beta2, eps, step), "csplat_adam_step")
        return None
