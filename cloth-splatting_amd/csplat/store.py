"""GaussianStore: capacity-based storage of everything that has one row per Gaussian.

The reference changes the number of Gaussians by building new tensors: every densification / pruning re-creates the seven
nn.Parameters with boolean-mask indexing or torch.cat, does the same to both Adam moments of each, deletes and re-inserts the
optimizer state entries and re-allocates the statistics (/root/reference/scene_reconstruction/gaussian_model.py:266-341,
gaussian_mesh.py:336-431).  Here the rows live in buffers with spare capacity (two sets, A and B):

  * the nn.Parameter OBJECTS, the optimizer's param_groups and its state dicts are never replaced -- after a surgery their
    `.data` / `exp_avg` / `exp_avg_sq` are re-pointed at the first P rows of the live buffer set (no allocation);
  * pruning = stream compaction from the live set into the other one and a swap: one prefix scan of the keep mask
    (csplat_mask_to_map) and ONE launch that moves the rows of all 7 attributes, their 14 moments and the 4 statistics
    (csplat_rows_scatter); order is preserved, so the result equals the reference's `tensor[mask]` row for row;
  * cloning / splitting = appending rows behind the live range of the same set (moments of new rows zero-filled by the same
    launch); capacity doubles when exceeded, which is the only time memory is allocated.

CUDA tensors go through the HIP kernels; tensors on any other device take the torch-op branch with the same semantics (the
CPU replay of the reference's own run, tests/golden/densify.npz, drives it)."""
import ctypes as C

import torch

GROUPS = (("face_bary", "face_bary"), ("face_offset", "face_offset"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"),
          ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation"))
STATS = ("face_ids", "pos_gradient_accum", "denom", "max_radii2D")


class GaussianStore:
    def __init__(self, owner, capacity=None):
        self.owner = owner
        self.P = int(owner.face_bary.shape[0])
        self.cap = 0
        self.sets = [None, None]
        self.live = 0
        self.allocations = 0          # (tests: surgery must not allocate while the capacity suffices)
        self._allocate(int(capacity) if capacity else max(2 * self.P, 1024), adopt=True)

    # ---- layout -------------------------------------------------------------------------------------------------------
    def _param(self, name):
        for group in self.owner.optimizer.param_groups:
            if group["name"] == name:
                return group["params"][0]
        raise KeyError(name)

    def _sources(self):
        """name -> current tensor of every row-wise array (attributes, their moments when Adam has state, statistics)"""
        o, out = self.owner, {}
        for name, attr in GROUPS:
            p = self._param(name)
            out["p:" + name] = p.data
            st = o.optimizer.state.get(p, None)
            if st is not None and "exp_avg" in st:
                out["m:" + name], out["v:" + name] = st["exp_avg"], st["exp_avg_sq"]
        for s in STATS:
            out[s] = getattr(o, s)
        return out

    def _allocate(self, cap, adopt=False):
        src = self._sources()
        new_sets = []
        for _ in range(2):
            new_sets.append({k: torch.zeros((cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device) for k, v in src.items()})
        for k, v in src.items():
            new_sets[0][k][:self.P].copy_(v[:self.P])
        self.sets, self.live, self.cap = new_sets, 0, cap
        self.allocations += 1
        self._rebind()

    def _track_new_state(self):
        """Adam creates its moments lazily at the first step(): adopt the ones that appeared since the buffers were laid out"""
        src = self._sources()
        missing = [k for k in src if k not in self.sets[0]]
        if not missing:
            return
        for k in missing:
            v = src[k]
            for s in self.sets:
                s[k] = torch.zeros((self.cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            self.sets[self.live][k][:self.P].copy_(v[:self.P])
        self._rebind()

    def _rebind(self):
        o, cur, P = self.owner, self.sets[self.live], self.P
        for name, attr in GROUPS:
            p = self._param(name)
            p.data = cur["p:" + name][:P]
            setattr(o, attr, p)
            st = o.optimizer.state.get(p, None)
            if st is not None and "m:" + name in cur:
                st["exp_avg"], st["exp_avg_sq"] = cur["m:" + name][:P], cur["v:" + name][:P]
        for s in STATS:
            setattr(o, s, cur[s][:P])
        if hasattr(o, "invalidate_caches"):
            o.invalidate_caches()

    def drop_grads(self, names=None):
        """the reference's surgery leaves fresh Parameters without .grad, so the optimizer step that follows skips them"""
        for name, _ in GROUPS:
            if names is None or name in names:
                self._param(name).grad = None

    def ensure_capacity(self, rows):
        if rows > self.cap:
            self._allocate(max(2 * self.cap, rows))

    # ---- the two primitives ---------------------------------------------------------------------------------------------
    def _mask_to_map(self, mask, base):
        """destination row of every source row (int32, -1 = dropped) and the number of kept rows"""
        mask = mask.reshape(-1).to(torch.bool)
        n = int(mask.shape[0])
        if mask.is_cuda:
            from . import native as _n
            m8 = mask.to(torch.uint8).contiguous()
            out = torch.empty(n, dtype=torch.int32, device=mask.device)
            cnt = torch.zeros(1, dtype=torch.int32, device=mask.device)
            tmp = torch.empty(int(_n.lib.csplat_mask_to_map_temp_bytes(n)), dtype=torch.uint8, device=mask.device)
            with _n.on_device(mask.device):
                _n.check(_n.lib.csplat_mask_to_map(_n.stream_handle(mask.device), n, _n.ptr(m8), int(base), _n.ptr(out), _n.ptr(cnt),
                                                   _n.ptr(tmp)), "csplat_mask_to_map")
            return out, int(cnt.item())
        ranks = torch.cumsum(mask.to(torch.int64), 0) - 1
        out = torch.where(mask, ranks + base, torch.full_like(ranks, -1)).to(torch.int32)
        return out, int(mask.sum())

    def _scatter(self, pairs, row_map, n_rows):
        """pairs: (src tensor [n_rows, ...] or None = zeros, dst buffer); dst[row_map[i]] = src[i] for row_map[i] >= 0"""
        if not pairs or n_rows == 0:
            return
        if pairs[0][1].is_cuda:
            from . import native as _n
            dev = pairs[0][1].device
            keep = [None if s is None else s.contiguous() for s, _ in pairs]
            for c0 in range(0, len(pairs), 32):
                chunk = pairs[c0:c0 + 32]
                n = len(chunk)
                srcs = (C.c_void_p * n)(*[None if k is None else k.data_ptr() for k in keep[c0:c0 + 32]])
                dsts = (C.c_void_p * n)(*[d.data_ptr() for _, d in chunk])
                rb = (C.c_int64 * n)(*[d[0].numel() * d.element_size() for _, d in chunk])
                with _n.on_device(dev):
                    _n.check(_n.lib.csplat_rows_scatter(_n.stream_handle(dev), n, C.cast(srcs, C.c_void_p), C.cast(dsts, C.c_void_p),
                                                        C.cast(rb, C.c_void_p), int(n_rows), _n.ptr(row_map)), "csplat_rows_scatter")
            return
        valid = row_map >= 0
        to = row_map[valid].long()
        for s, d in pairs:
            d[to] = 0 if s is None else s[:n_rows][valid]

    # ---- surgery --------------------------------------------------------------------------------------------------------
    def compact(self, keep_mask):
        """keep the rows with keep_mask set, order preserved: live set -> other set, swap"""
        self._track_new_state()
        cur, other = self.sets[self.live], self.sets[1 - self.live]
        row_map, count = self._mask_to_map(keep_mask, 0)
        self._scatter([(cur[k][:self.P], other[k]) for k in cur], row_map, self.P)
        self.live, self.P = 1 - self.live, count
        self._rebind()
        self.drop_grads()

    def append_selected(self, select_mask):
        """clone the selected rows behind the live range (attributes and face ids copied, moments zero)"""
        self._track_new_state()
        n_new = int(select_mask.sum())
        self.ensure_capacity(self.P + n_new)
        cur = self.sets[self.live]
        row_map, count = self._mask_to_map(select_mask, self.P)
        pairs = []
        for k in cur:
            if k.startswith("p:") or k == "face_ids":
                pairs.append((cur[k][:self.P], cur[k]))
            elif k.startswith("m:") or k.startswith("v:"):
                pairs.append((None, cur[k]))
        self._scatter(pairs, row_map, self.P)
        self.P += count
        self._rebind()
        self.drop_grads()

    def append_rows(self, values, face_ids):
        """append explicit rows: values = {group name: tensor [n, ...]}, face_ids [n]; moments of the new rows are zero"""
        self._track_new_state()
        n_new = int(face_ids.shape[0])
        self.ensure_capacity(self.P + n_new)
        cur = self.sets[self.live]
        if n_new:
            row_map = torch.arange(self.P, self.P + n_new, dtype=torch.int32, device=face_ids.device)
            pairs = [(values[name].to(cur["p:" + name].dtype), cur["p:" + name]) for name, _ in GROUPS]
            pairs.append((face_ids.to(cur["face_ids"].dtype), cur["face_ids"]))
            pairs += [(None, cur[k]) for k in cur if k.startswith("m:") or k.startswith("v:")]
            self._scatter(pairs, row_map, n_new)
        self.P += n_new
        self._rebind()
        self.drop_grads()

    def reset_stats(self):
        cur = self.sets[self.live]
        for s in ("pos_gradient_accum", "denom", "max_radii2D"):
            cur[s][:self.P].zero_()
