"""View-parallel execution over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  The reference is single-GPU; this is the new
capability described in SURVEY.md 5.8 / 8(e):

  * Gaussians, mesh and simulator are replicated; the cameras of a step are dealt round-robin over the ranks;
  * the gradients of ALL parameters live in ONE persistent flat fp32 buffer (FlatGrads): every `p.grad` is a view into it, so
    autograd accumulates straight into the buffer and the step's single all-reduce(sum) needs no gather / scatter copies
    (240 B per Gaussian + simulator gradients + a small tail for the step's scalars and the screen-space statistics --
    xGMI is point-to-point, one large message per step is the per-link-friendly shape; no bucketing is needed at 24-55 MB);
  * densification statistics are reduced with the operator that keeps replicas bit-identical:
    screen-space gradient (part of the flat buffer) -> sum, max_radii2D -> max (visibility = radius > 0).
"""
import os
import time

import torch
import torch.distributed as dist

# CSPLAT_FORCE_DIST=1: treat an initialised process group of ONE rank as distributed -- every collective of the view-parallel step is
# then really issued (RCCL on a 1-GPU box: library load, communicator, the all-reduce started from the autograd hook and its stream
# ordering), where the product would otherwise skip them as the identity they are.  Tests and bench.py's 1-rank `collective` leg.
FORCE_DIST = os.environ.get("CSPLAT_FORCE_DIST", "") == "1"


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_DIST)


def world_rank():
    return (dist.get_world_size(), dist.get_rank()) if is_dist() else (1, 0)


def shard_indices(n, rank=None, world=None):
    """indices i < n with i mod world == rank; ranks beyond n get an empty list and still join the collectives."""
    w, r = world_rank()
    rank = r if rank is None else rank
    world = w if world is None else world
    return [i for i in range(n) if i % world == rank]


def shard_views(views, rank=None, world=None):
    views = list(views)
    return [views[i] for i in shard_indices(len(views), rank, world)]


class FlatGrads:
    """One flat fp32 buffer that ends a step holding the gradient of every one of `params` in its own slice (+ `extra` trailing floats
    for whatever else has to be summed over ranks in the same message), and the `.grad` tensors are VIEWS of those slices.

    How a gradient gets there (round 4): bind() registers every parameter's slice as its gradient SINK (csplat.native.GRAD_SINK) and
    clears `.grad`.  The autograd nodes that finish a parameter's gradient -- the rasterizer's K8 for its direct inputs, the activation
    and mesh-transform adjoints, the simulator's backward -- write it straight into the slice and return a fresh view of it, which
    autograd ADOPTS as `.grad` (AccumulateGrad steals a gradient nobody else holds): no temporary, no zero fill of the slice, no
    in-place add.  A gradient that arrives any other way (a node without sink support, a sum of two contributions) is copied into the
    slice by the post-accumulate hook.  Slices nothing wrote are zero: the first step of a step SHAPE zero-fills the whole buffer and
    learns which parameters were written in place; afterwards only the other slices and the tail are cleared.
    Rebuilt by the owner when the parameter set changes shape (densification).

    `early` = how many LEADING parameters form the early bucket: their gradients are complete long before the backward pass ends
    (the Gaussian parameters: final when the rasterizer's K8 and the activation / mesh-transform adjoints have run, while the
    simulator MLP and the regularisers are still in backward).  The post-accumulate hook of the LAST of them to receive its gradient
    starts the all-reduce of that slice right away (async: RCCL runs it on its own stream behind the kernels queued so far), so the
    exchange of ~95 % of the Gaussian bytes overlaps the rest of backward; all_reduce() then sends the remainder (simulator gradients +
    tail) and waits for both.  The sums are the same sums: results equal the one-shot all-reduce bit for bit."""

    def __init__(self, params, extra=0, early=0):
        import weakref
        self.params = [p for p in params]
        assert self.params, "FlatGrads: no parameters"
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == self.params[0].dtype for p in self.params), "FlatGrads: mixed devices / dtypes"
        self.shapes = [tuple(p.shape) for p in self.params]
        self.sizes = [p.numel() for p in self.params]
        # every slice starts on a 256-byte boundary (kernels that write a gradient in place use 16-byte stores); the padding is zero
        self.offsets, o = [], 0
        for n in self.sizes:
            self.offsets.append(o)
            o += (n + 63) & ~63
        self.n_param = o
        self.extra = int(extra)
        self.early = int(early)
        self.n_early = self.offsets[self.early] if self.early < len(self.sizes) else self.n_param
        # (padded to a multiple of 64 floats per rank: the direct exchange cuts the buffer into one chunk per rank; the padding is zero)
        w_ = world_rank()[0]
        self.n_used = self.n_param + self.extra
        self.flat_all = torch.zeros(-(-self.n_used // (64 * w_)) * 64 * w_, dtype=self.params[0].dtype, device=dev)
        self.flat = self.flat_all[:self.n_used]
        self.views = [self.flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.sizes, self.shapes)]
        self._refs = [weakref.ref(p) for p in self.params]
        self.tail = self.flat[self.n_param:]
        self.last_allreduce_ms = 0.0
        self.algo = os.environ.get("CSPLAT_ALLREDUCE", "rccl") if os.environ.get("CSPLAT_ALLREDUCE", "rccl") in ("rccl", "direct") else "rccl"
        self.algo_note = None
        self._slice_work, self._sliced = [], []
        self.early_fired = 0           # how many steps sent their early slice from the backward hook (tests, bench)
        self.in_place = 0              # gradients adopted in place (written into the slice by their last kernel) since creation
        self.copied = 0                # ... and gradients that had to be copied into their slice
        self._touched = [False] * len(self.params)
        self._placed = [False] * len(self.params)
        self._union = {}
        # per step SHAPE (the `key` of bind(): static / dynamic stage, number of cameras -- whatever changes the graph), learned on the
        # first step of each shape: which early parameters receive a gradient, and which parameters' slices are written in place
        self._early_expect = {}
        self._in_place_expect = {}
        self._key = None
        self._early_left = None
        self._early_work = None
        self._early_late = []          # early parameters that received a gradient AFTER the early slice had left (must stay empty)
        self._group = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._mark(i)) for i, p in enumerate(self.params)]

    def _mark(self, i):
        def hook(p):
            self._touched[i] = True
            g = p.grad
            if g is not None and g.dtype == self.flat.dtype and g.data_ptr() == self.views[i].data_ptr() and g.is_contiguous():
                self.in_place += 1
                self._placed[i] = True
            elif g is not None:            # arrived some other way: into the slice, and `.grad` becomes the slice
                with torch.no_grad():
                    self.views[i].copy_(g)
                p.grad = self.views[i]
                self.copied += 1
            if i < self.early and self._early_work is not None:
                self._early_late.append(i)
            if self._early_left is not None and i in self._early_left:
                self._early_left.discard(i)
                if not self._early_left:
                    self._start_early()
        return hook

    def _start_early(self):
        """every early parameter that gets a gradient has it: send the early slice now, under the rest of backward"""
        self._early_left = None
        if is_dist() and self.n_early > 0:
            self.early_fired += 1
            self._early_work = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, group=self._group, async_op=True)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.unbind()

    def matches(self, params, extra=0, early=0):
        params = list(params)
        return len(params) == len(self.params) and int(extra) == self.extra and int(early) == self.early and \
            all(a is b and tuple(a.shape) == s for a, b, s in zip(params, self.params, self.shapes))

    def bind(self, group=None, key=None):
        from . import native as _n
        self._touched = [False] * len(self.params)
        self._placed = [False] * len(self.params)
        self._early_work = None
        self._early_late = []
        self._group = group
        self._key = key
        known = self._in_place_expect.get(key)
        if known is None:
            self.flat.zero_()                      # first step of this shape: everything (and learn who writes in place)
        else:
            for i, hit in enumerate(known):        # slices their last kernel overwrites need no clearing; the others and the tail do
                if not hit:
                    self.views[i].zero_()
            if self.extra:
                self.tail.zero_()
        # the early bucket fires when the same early parameters as in the previous step OF THIS SHAPE have their gradients (the graph
        # of a step shape is static); the first step of a shape learns the set and sends everything at the end
        expect = self._early_expect.get(key)
        self._early_left = set(expect) if (self.early and expect) else None
        for i, p in enumerate(self.params):
            p.grad = None
            if self.flat.is_cuda:
                _n.GRAD_SINK[id(p)] = (self._refs[i], self.flat, self.offsets[i], self.shapes[i], [False])     # ([used]: one writer per bind)

    def unbind(self):
        """the sinks are for ONE step: an ordinary (one-rank) step that follows must not write into this buffer"""
        from . import native as _n
        for p in self.params:
            e = _n.GRAD_SINK.get(id(p))
            if e is not None and e[1] is self.flat:
                del _n.GRAD_SINK[id(p)]

    def drop_untouched(self, key=None, group=None):
        """after backward + all_reduce: `p.grad` = its slice for every parameter that SOME rank's backward wrote (a rank without a
        camera got no gradient of its own for the Gaussians, but holds the sum now), None for the others -- a parameter outside the
        graph (`face_offset`, a frozen group) must end the step with grad None exactly as in the one-rank step, or Adam would create
        state for it and advance its step count.  The union over ranks is one tiny all-reduce(max) + host read the first time a step
        shape `key` is seen (the graph of a step is static: same cameras per rank, same parameters in it) and cached afterwards."""
        if key not in self._union:
            t = torch.tensor([int(b) for b in self._touched], dtype=torch.int32, device=self.flat.device)
            if is_dist():
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            self._union[key] = [bool(b) for b in t.tolist()]
        for i, (p, hit) in enumerate(zip(self.params, self._union[key])):
            p.grad = self.views[i] if hit else None

    def all_reduce(self, group=None, timed=False):
        """sum over ranks, in place: the part of the buffer the early bucket has not already sent, then wait for both.  timed=True
        brackets the call with device synchronisation and records its wall time -- `last_allreduce_ms` = what this call took with
        the device drained first, i.e. the EXPOSED time of the step's exchange when the early bucket was in flight (bench.py reports
        the one-shot time next to it); the default leaves it asynchronous with respect to the host."""
        self.unbind()
        if self._early_late:
            # the slice that left from the hook did not hold this gradient yet: the sums the other ranks are about to use are wrong.
            # A step whose graph differs from the one its `key` was learned on -- give bind() a key that tells them apart.
            late, self._early_late = self._early_late, []
            raise RuntimeError(f"FlatGrads: early parameter(s) {sorted(set(late))} received a gradient after the early slice had been "
                               f"sent (step shape {self._key!r}): the step's graph changed under one bind() key")
        known = self._in_place_expect.get(self._key)
        if known is None:
            self._in_place_expect[self._key] = list(self._placed)
        else:
            # a slice that was left uncleared because its gradient is written in place -- and then was not: it holds the previous
            # step's sum.  Clear it now (possible unless the early slice is already on its way) and forget the expectation.
            stale = [i for i, hit in enumerate(known) if hit and not self._placed[i]]
            if stale:
                if self._early_work is not None and any(i < self.early for i in stale):
                    raise RuntimeError(f"FlatGrads: parameter(s) {stale} were expected to be written in place (step shape {self._key!r}) and "
                                       "were not, after the early slice had been sent")
                for i in stale:
                    if not self._touched[i]:
                        self.views[i].zero_()
                self._in_place_expect[self._key] = [a and b for a, b in zip(known, self._placed)]
        if self.early and self._key not in self._early_expect:      # learned once per step shape: which early parameters it touches
            self._early_expect[self._key] = [i for i in range(self.early) if self._touched[i]]
        elif self.early and self._early_work is None and self._early_expect[self._key] and \
                not all(self._touched[i] for i in self._early_expect[self._key]):
            self._early_expect[self._key] = [i for i in range(self.early) if self._touched[i]]      # (the hook never fired: relearn)
        if not is_dist():
            return self.flat
        if timed and self.flat.is_cuda:
            torch.cuda.synchronize(self.flat.device)
            t0 = time.perf_counter()
        # the protocol is the same on every rank whatever its share of the cameras: with an early bucket, ALWAYS two collectives in
        # this order -- the early slice (already in flight where the hook fired; sent now where it did not: the first step of a buffer,
        # a rank without a camera) and then the rest
        self._early_left = None
        if self.early and self.n_early > 0:
            if self._early_work is None:
                self._early_work = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, group=group, async_op=True)
            rest = self.flat[self.n_early:]
        else:
            rest = self.flat
        if rest.numel():
            if rest.data_ptr() == self.flat.data_ptr() and rest.numel() == self.flat.numel():
                self._exchange(group)           # (the whole buffer in one message: the tuned algorithm)
            else:
                dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=group)
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None
        if timed and self.flat.is_cuda:
            torch.cuda.synchronize(self.flat.device)
            self.last_allreduce_ms = (time.perf_counter() - t0) * 1e3
        return self.flat


    # ---- the exchange itself.  ALGO: "rccl" = one all-reduce of the buffer (the library's ring / tree); "direct" = reduce-scatter by
    # all-to-all + local sum + all-gather: xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU,
    # /opt/skills/guides/MI355X_MICROARCH.md), and an all-to-all drives all seven links of every GPU at once with 1/world of the buffer
    # each way, where a ring pushes 2 (w - 1) / w of the buffer through per-link hops; every chunk is summed by ONE rank in rank order and
    # broadcast, so the replicas stay bit-identical.  Which one is faster on a given node is MEASURED, not guessed: `tune_exchange()`
    # times both on this buffer and keeps the faster (the ranks agree through a MAX all-reduce of the timings).
    def _exchange(self, group=None):
        if self.algo == "direct":
            try:
                self._direct(group)
                return
            except (RuntimeError, NotImplementedError, ValueError) as e:      # (a backend without all-to-all: gloo on device tensors)
                self.algo, self.algo_note = "rccl", "direct exchange unavailable: " + repr(e)[:100]
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def _direct(self, group=None):
        w, r = dist.get_world_size(group), dist.get_rank(group)
        chunk = self.flat_all.numel() // w
        send = self.flat_all.view(w, chunk)
        if getattr(self, "_recv", None) is None or self._recv.shape != send.shape:
            self._recv = torch.empty_like(send)
        dist.all_to_all_single(self._recv.view(-1), self.flat_all, group=group)          # row j = rank j's copy of MY chunk
        torch.sum(self._recv, dim=0, out=send[r])                                      # fixed order over ranks
        dist.all_gather_into_tensor(self.flat_all, send[r], group=group)                 # (in place: input = output chunk r)

    def tune_exchange(self, group=None, reps=5):
        """time both exchange algorithms on this buffer (contents are summed `reps` times each: call it on a buffer whose contents do not
        matter -- before the first step) and keep the faster; returns {"rccl_ms", "direct_ms", "chosen"}"""
        if not is_dist() or not self.flat.is_cuda:
            return {"rccl_ms": None, "direct_ms": None, "chosen": self.algo}
        res = {}
        for algo in ("rccl", "direct"):
            self.algo = algo
            try:
                self._exchange(group)
                torch.cuda.synchronize(self.flat.device)
                if self.algo != algo:
                    res[algo] = float("inf")
                    continue
                dist.barrier(group)
                torch.cuda.synchronize(self.flat.device)
                t0 = time.perf_counter()
                for _ in range(reps):
                    self._exchange(group)
                torch.cuda.synchronize(self.flat.device)
                res[algo] = (time.perf_counter() - t0) / reps * 1e3
            except Exception:
                res[algo] = float("inf")
        t = torch.tensor([res["rccl"], res["direct"]], dtype=torch.float64, device=self.flat.device)
        t = torch.nan_to_num(t, posinf=1e9)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        rc, dr = float(t[0]), float(t[1])
        self.algo = "direct" if dr < rc else "rccl"
        self.flat_all.zero_()
        return {"rccl_ms": None if rc >= 1e9 else round(rc, 4), "direct_ms": None if dr >= 1e9 else round(dr, 4), "chosen": self.algo}

    # ---- the exchange in SLICES of Gaussian rows (round 6): K8 finishes the gradient rows of a range of Gaussians per launch
    # (csplat_backward_views_parts); the rows of slice g travel while slice g + 1 computes
    def per_gaussian(self, P):
        """indices of the parameters whose leading dimension is the Gaussian count P (their gradients are finished row range by row range)"""
        return [i for i, s_ in enumerate(self.shapes) if len(s_) >= 1 and s_[0] == P and self.sizes[i] % max(P, 1) == 0]

    def slice_ranges(self, P, row_lo, row_hi, which=None):
        """the flat sub-ranges holding rows [row_lo, row_hi) of every per-Gaussian parameter"""
        out = []
        for i in (self.per_gaussian(P) if which is None else which):
            wdt = self.sizes[i] // P
            if row_hi > row_lo:
                out.append(self.flat[self.offsets[i] + row_lo * wdt:self.offsets[i] + row_hi * wdt])
        return out

    def start_ranges(self, ranges, group=None):
        """all-reduce(sum) of a list of flat sub-ranges, asynchronous: queued on the collective's stream behind the kernels launched so
        far; the handles are waited for in finish_sliced().  One grouped launch where the backend coalesces (RCCL), else one per range."""
        if not is_dist() or not ranges:
            return
        self.unbind()
        if dist.get_backend(group) == "nccl" and len(ranges) > 1 and hasattr(dist, "_coalescing_manager"):
            # (decided up front, never as a fallback: a manager that failed half way would have issued some of the sums already)
            with dist._coalescing_manager(group=group, device=self.flat.device, async_ops=True) as cm:
                for t in ranges:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            self._slice_work.append(cm)
        else:
            for t in ranges:
                self._slice_work.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
        self._sliced.extend((int(t.data_ptr()), int(t.numel())) for t in ranges)

    def finish_sliced(self, P, group=None):
        """everything the slices did not carry (parameters that are not per-Gaussian, the tail) in one more collective, then wait for all.
        The slices must have covered every row of every per-Gaussian parameter (checked): the sums equal the one-shot all-reduce's."""
        self.unbind()
        if not is_dist():
            self._slice_work, self._sliced = [], []
            return self.flat
        covered = {}
        base = int(self.flat.data_ptr())
        for ptr, n in self._sliced:
            covered[(ptr - base) // 4] = n
        pg = set(self.per_gaussian(P))
        rest = []
        for i in range(len(self.params)):
            if i in pg:
                o, end = self.offsets[i], self.offsets[i] + self.sizes[i]
                while o < end:
                    n = covered.get(o)
                    if not n:
                        raise RuntimeError(f"FlatGrads.finish_sliced: rows of parameter {i} at flat offset {o} were sent by no slice")
                    o += n
            else:
                rest.append(self.flat[self.offsets[i]:self.offsets[i] + self.sizes[i]])
        if self.extra:
            rest.append(self.tail)
        if rest:
            self.start_ranges(rest, group)
        for wk in self._slice_work:
            wk.wait()
        self._slice_work, self._sliced = [], []
        return self.flat


def flat_grads_for(owner, params, extra=0, early=0):
    """the FlatGrads cached on `owner` for exactly these parameter objects (rebuilt when they were replaced or resized)"""
    params = list(params)
    fg = owner.__dict__.get("_flat_grads") if hasattr(owner, "__dict__") else None
    if fg is None or not fg.matches(params, extra, early):
        if fg is not None:
            fg.close()
        fg = FlatGrads(params, extra, early)
        try:
            owner._flat_grads = fg
        except Exception:
            pass
    return fg


def allreduce_flat(tensors, op=None, group=None):
    """All-reduce a list of same-dtype tensors as ONE flat buffer, in place (copying variant for ad-hoc tensor lists; the
    training step uses FlatGrads, which needs no copies).  Returns the flat buffer."""
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return None
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if is_dist():
        dist.all_reduce(flat, op=op or dist.ReduceOp.SUM, group=group)
    pieces, o = [], 0
    for t in tensors:
        n = t.numel()
        pieces.append(flat[o:o + n].view_as(t))
        o += n
    try:
        torch._foreach_copy_(tensors, pieces)        # one multi-tensor launch instead of one copy per parameter
    except (AttributeError, RuntimeError):
        for t, src in zip(tensors, pieces):
            t.copy_(src)
    return flat


def allreduce_gradients(params, group=None):
    """sum the .grad of every parameter over ranks (parameters without grad contribute zeros so that all ranks
    issue the same collective)."""
    grads = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    return allreduce_flat(grads, group=group)


def reduce_max_radii(radii, group=None):
    """train_utils.py:276-277: the largest screen radius of every Gaussian over the step's cameras -> max over ranks."""
    if is_dist():
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    return radii


def reduce_densification_stats(viewspace_grad, radii, visibility, group=None):
    """train_utils.py:276-277,290-292: viewspace gradient (sum over cams -> sum over ranks), radii (max), visibility (any)."""
    if not is_dist():
        return viewspace_grad, radii, visibility
    dist.all_reduce(viewspace_grad, op=dist.ReduceOp.SUM, group=group)
    packed = torch.stack([radii.to(torch.int32), visibility.to(torch.int32)])
    dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    return viewspace_grad, packed[0].to(radii.dtype), packed[1].to(torch.bool)
