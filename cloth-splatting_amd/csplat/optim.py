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
        # (torch returns the per-parameter dicts BY REFERENCE: sd["state"][i] is self.state[p].  They are rebuilt here, never edited --
        #  replacing a live moment by its clone would detach the optimizer from the store's buffers until the next compaction)
        over = lambda v: torch.is_tensor(v) and v.numel() and v.untyped_storage().nbytes() > v.numel() * v.element_size()  # noqa: E731
        sd["state"] = {i: {k: (v.clone() if over(v) else v) for k, v in st.items()} for i, st in sd["state"].items()}
        return sd

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not self._fusable():
            self.__dict__.pop("_step_cache", None)
            return super().step(closure)
        return self._fused_step()

    @torch.no_grad()
    def step_now(self):
        """step() minus torch.optim's per-call wrapper (profiler record, hook dispatch: ~35 us of host time per call on a step that is
        bound by the host) -- what csplat.train.train_step calls.  With step hooks registered, or anything the kernel does not cover,
        this IS step()."""
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _global_step_hooks() or not self._fusable():
            return self.step()
        return self._fused_step()

    # ---- the step as a launch that depends on NO per-step host value (csplat_adam_step_dev): for a training step recorded into a hipGraph
    def captured_setup(self):
        """(Re-)creates the device-side state a captured step reads: the step count (int32 [1], from the host-side counters, which must
        agree over all parameters that have state) and the learning rates (double, one per tensor with state, in param_groups order).
        Call it outside the capture, after at least one ordinary step() (the moments must exist)."""
        items, counts = [], set()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                st = self.state.get(p, {})
                if "exp_avg" not in st:
                    continue
                items.append((p, st, float(beta1), float(beta2), float(group["eps"]), group))
                counts.add(int(st["step"].item()))
        if not items:
            raise RuntimeError("GroupedAdam.captured_setup: no optimizer state yet (take one ordinary step first)")
        if len(counts) != 1 or len({it[2:5] for it in items}) != 1:
            raise RuntimeError("GroupedAdam.captured_setup: the tensors must share one step count and one (beta1, beta2, eps)")
        dev = items[0][0].device
        count = counts.pop()
        cap = self.__dict__.get("_cap")
        if cap is not None and cap["lr"].numel() == len(items) and cap["state"].device == dev:
            # the device words are allocated ONCE per optimizer and refreshed in place: a hipGraph recorded earlier (another step shape
            # of csplat.train.CapturedStep) has their addresses baked in and keeps reading live values
            cap["items"], cap["lr_seen"] = items, None
            cap["state"].fill_(count)
            cap["dev_step"] = count
        else:
            self._cap = {"items": items, "state": torch.tensor([count], dtype=torch.int32, device=dev),
                         "lr": torch.zeros(len(items), dtype=torch.float64, device=dev),
                         "lr_host": torch.zeros(len(items), dtype=torch.float64).pin_memory(), "lr_seen": None, "dev_step": count,
                         # bumped whenever the device words are RE-allocated: recordings that hold the old addresses are stale
                         "epoch": (cap["epoch"] + 1) if cap is not None else 0}
        self.captured_refresh_lr()
        return self._cap

    def captured_refresh_lr(self):
        """Before a replay: the groups' current learning rates -> the device table (a schedule edits param_groups[i]['lr'] on the host;
        no-op when unchanged), and the device-side step count re-synchronised with the host's counters when an ORDINARY step ran in
        between (csplat.train.CapturedStep runs every 1000th iteration eagerly: the recorded Adam launch would otherwise use a stale
        count -- wrong bias correction -- and the replay's sequence word would never match)."""
        cap = self._cap
        lrs = tuple(float(it[5]["lr"]) for it in cap["items"])
        if lrs != cap["lr_seen"]:
            cap["lr_host"].copy_(torch.tensor(lrs, dtype=torch.float64))
            cap["lr"].copy_(cap["lr_host"], non_blocking=True)
            cap["lr_seen"] = lrs
        host_step = int(cap["items"][0][1]["step"].item())          # (a CPU tensor)
        if host_step != cap["dev_step"]:
            cap["state"].fill_(host_step)
            cap["dev_step"] = host_step

    def step_captured(self, valid):
        """the launch itself (record it under stream capture): every tensor of captured_setup() that has a gradient NOW; `valid` = a
        uint32 device word (0 -> nothing is applied, the count does not advance)"""
        cap = self._cap
        items = [it for it in cap["items"] if it[0].grad is not None]
        if len(items) != len(cap["items"]):
            raise RuntimeError("GroupedAdam.step_captured: a tensor with optimizer state has no gradient in the captured step")
        n = len(items)
        ptrs = [(C.c_void_p * n)(*[t.data_ptr() for t in ts]) for ts in ([it[0] for it in items], [it[0].grad for it in items],
                                                                          [it[1]["exp_avg"] for it in items], [it[1]["exp_avg_sq"] for it in items])]
        for it in items:
            g = it[0].grad
            if not (g.is_contiguous() and g.dtype == torch.float32 and it[0].is_contiguous() and it[0].dtype == torch.float32):
                raise RuntimeError("GroupedAdam.step_captured: fp32 contiguous parameters and gradients only")
        numel = (C.c_int64 * n)(*[it[0].numel() for it in items])
        dev = items[0][0].device
        with _n.on_device(dev):
            _n.check(_n.lib.csplat_adam_step_dev(_n.stream_handle(dev), n, C.cast(ptrs[0], C.c_void_p), C.cast(ptrs[1], C.c_void_p),
                                                 C.cast(ptrs[2], C.c_void_p), C.cast(ptrs[3], C.c_void_p), C.cast(numel, C.c_void_p),
                                                 _n.ptr(cap["lr"]), items[0][2], items[0][3], items[0][4], _n.ptr(cap["state"]),
                                                 _n.ptr(valid)), "csplat_adam_step_dev")

    def captured_advance_host(self):
        """after a replay whose valid word was 1: the host-side step counters follow the device's"""
        steps = [it[1]["step"] for it in self._cap["items"]]
        torch._foreach_add_(steps, 1)
        self._cap["dev_step"] += 1
        self.__dict__.pop("_step_cache", None)

    def zero_grad_now(self):
        """zero_grad(set_to_none=True) without the wrapper's bookkeeping"""
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def _fused_step(self):
        buckets = {}
        steps = []
        state = self.state
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            bkey = (float(beta1), float(beta2), float(group["eps"]))
            lr = float(group["lr"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not g.is_contiguous():
                    g = g.contiguous()
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32):
                    raise RuntimeError("GroupedAdam: optimizer state must be contiguous fp32")
                steps.append(st["step"])
                buckets.setdefault((p.device,) + bkey, []).append((p, g, m, v, lr, st))
        if not steps:
            return None
        # the step counts AFTER this step as Python ints.  torch keeps them as CPU tensors in the state (that is what state_dict()
        # saves) and reading them back costs more than the rest of this function.  Usual case: the same state tensors as last time,
        # written by nobody but our own _foreach_add_ -> every count is last time's + 1 (ONE identity / version check for all)
        sig = (tuple(map(id, steps)), sum(t._version for t in steps))
        cache = self.__dict__.get("_step_cache")
        if cache is not None and cache[0] == sig[0] and cache[1] == sig[1] and all(a is b for a, b in zip(cache[3], steps)):
            counts = [n + 1 for n in cache[2]]
        else:
            counts = [int(t.item()) + 1 for t in steps]
        torch._foreach_add_(steps, 1)              # the state's own counters, all in one call
        self._step_cache = (sig[0], sig[1] + len(steps), counts, steps)
        count_of = {id(t): n for t, n in zip(steps, counts)}
        for (dev, beta1, beta2, eps), all_items in buckets.items():
            by_step = {}
            for it in all_items:
                by_step.setdefault(count_of[id(it[5]["step"])], []).append(it)
            for step, items in by_step.items():
                n = len(items)
                ptrs = [(C.c_void_p * n)(*[it[k].data_ptr() for it in items]) for k in range(4)]
                numel = (C.c_int64 * n)(*[it[0].numel() for it in items])
                lrs = (C.c_double * n)(*[it[4] for it in items])
                with _n.on_device(dev):
                    _n.check(_n.lib.csplat_adam_step(_n.stream_handle(dev), n, C.cast(ptrs[0], C.c_void_p), C.cast(ptrs[1], C.c_void_p),
                                                     C.cast(ptrs[2], C.c_void_p), C.cast(ptrs[3], C.c_void_p), C.cast(numel, C.c_void_p),
                                                     C.cast(lrs, C.c_void_p), beta1, beta2, eps, step), "csplat_adam_step")
        return None


def _global_step_hooks():
    try:
        from torch.optim import optimizer as _o
        return bool(_o._global_optimizer_pre_hooks) or bool(_o._global_optimizer_post_hooks)
    except Exception:
        return True
