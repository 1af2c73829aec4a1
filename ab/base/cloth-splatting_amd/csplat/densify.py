"""Densification / pruning of mesh-anchored Gaussians on capacity-based storage.

What it computes is the reference's: /root/reference/scene_reconstruction/gaussian_mesh.py:336-431 (which Gaussians are
cloned / split / pruned, where the new ones go, how their attributes are derived) and gaussian_model.py:214-218, 408-431
(opacity reset, thresholds, statistics), on the schedule of train_utils.py:324-345.  HOW the rows move is not the reference's:
its Adam-state surgery (gaussian_model.py:266-341: new nn.Parameters, masked / concatenated copies of both moments, optimizer
state entries deleted and re-inserted) is replaced by csplat/store.py -- rows are compacted or appended inside capacity
buffers by two HIP entry points and the Parameter objects are re-pointed, never re-created.  Row order after every operation
equals the reference's (`tensor[mask]` / `torch.cat` order), so its own run is replayed bit for bit (tests/golden/densify.npz:
parameters, both Adam moments, step counters, face ids, statistics after every stage; on CPU tensors and on the GPU).

Multi-GPU note: every decision below is a pure function of (parameters, accumulated statistics, the RNG stream), and
csplat.dist makes the statistics identical on all ranks; with the same seed the replicas stay bit-identical through
densification without exchanging the new Gaussians."""
import torch

from .store import GaussianStore


def compute_barycentric_coordinates(points, triangles):
    """meshnet/data_utils.py:494-530: (u, v, w) of `points [n,3]` in `triangles [n,3,3]` (projection onto the plane)."""
    A, B, Cc = triangles[:, 0, :], triangles[:, 1, :], triangles[:, 2, :]
    AB, AC, AP = B - A, Cc - A, points - A
    dot00, dot01, dot02 = (AC * AC).sum(1), (AC * AB).sum(1), (AC * AP).sum(1)
    dot11, dot12 = (AB * AB).sum(1), (AB * AP).sum(1)
    denom = dot00 * dot11 - dot01 * dot01
    v = (dot11 * dot02 - dot01 * dot12) / denom
    w = (dot00 * dot12 - dot01 * dot02) / denom
    return torch.stack([1.0 - v - w, v, w], dim=1)


class DensifyMixin:
    percent_dense = 0.01

    def invalidate_caches(self):      # MeshGaussians overrides; the golden replay drives the mixin on a bare object
        pass

    # ---- statistics (gaussian_mesh.py:121-124, gaussian_model.py:427-430, train_utils.py:326-328) -------------------
    def densification_setup(self, percent_dense=0.01):
        P, dev = self.face_bary.shape[0], self.face_bary.device
        self.percent_dense = percent_dense
        self.pos_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.pos_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    # ---- storage ------------------------------------------------------------------------------------------------------
    @property
    def store(self):
        """the capacity buffers behind the parameters (created at the first surgery: 2x the current size)"""
        st = self.__dict__.get("_store")
        if st is None or st.owner is not self or st.P != self.face_bary.shape[0] or \
                st._param("face_bary").data_ptr() != st.sets[st.live]["p:face_bary"].data_ptr():
            if not hasattr(self, "pos_gradient_accum") or self.pos_gradient_accum.shape[0] != self.face_bary.shape[0]:
                self.densification_setup(self.percent_dense)
            st = GaussianStore(self)
            self._store = st
        return st

    def reset_opacity(self):
        """gaussian_model.py:214-217: opacities capped at 0.01, both Adam moments of the group zeroed, its step count kept"""
        from .gaussians import inverse_sigmoid
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity.data.copy_(new)
        state = self.optimizer.state.get(self._opacity, None)
        if state is not None and "exp_avg" in state:
            state["exp_avg"].zero_()
            state["exp_avg_sq"].zero_()
        self._opacity.grad = None      # (upstream swaps in a fresh Parameter: the optimizer step that follows skips the group)

    # ---- which rows go where (gaussian_mesh.py:336-431) ---------------------------------------------------------------------
    def prune_points(self, mask):
        """drop the Gaussians with mask set (attributes, moments, face ids, statistics), order preserved"""
        self.store.compact(~mask)

    def densification_postfix(self, new_face_bary, new_face_offset, new_face_ids, new_features_dc, new_features_rest,
                              new_opacities, new_scaling, new_rotation):
        """append the given Gaussians; statistics restart from zero for ALL rows (gaussian_mesh.py:359-361)"""
        st = self.store
        st.append_rows({"face_bary": new_face_bary, "face_offset": new_face_offset, "f_dc": new_features_dc, "f_rest": new_features_rest,
                        "opacity": new_opacities, "scaling": new_scaling, "rotation": new_rotation}, new_face_ids)
        st.reset_stats()

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        """large Gaussians with a large screen-space gradient are replaced by N samples of themselves (scale / (0.8 N)),
        re-anchored on their face by barycentric coordinates of the sampled position"""
        from .gaussians import build_rotation
        n_init, dev = self.face_bary.shape[0], self.face_bary.device
        padded = torch.zeros((n_init,), device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        scale_sel = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((scale_sel.size(0), 3), device=dev), std=scale_sel)    # (global RNG, as upstream)
        jitter = torch.bmm(build_rotation(self._rotation[sel]).repeat(N, 1, 1), samples.unsqueeze(-1)).squeeze(-1)
        new_xyz = self.get_xyz()[sel].repeat(N, 1) + jitter
        corners = self.mesh.pos[self.mesh.face[:, self.face_ids[sel]]].transpose(0, 1).repeat(N, 1, 1)
        self.densification_postfix(
            compute_barycentric_coordinates(new_xyz, corners), self.face_offset[sel].repeat(N, 1), self.face_ids[sel].repeat(N),
            self._features_dc[sel].repeat(N, 1, 1), self._features_rest[sel].repeat(N, 1, 1), self._opacity[sel].repeat(N, 1),
            torch.log(scale_sel / (0.8 * N)), self._rotation[sel].repeat(N, 1))
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool))))

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """small Gaussians with a large screen-space gradient are duplicated in place (one launch appends all attributes)"""
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        st = self.store
        st.append_selected(sel)
        st.reset_stats()

    # ---- gaussian_mesh.py:267-322 -------------------------------------------------------------------------------------
    def _face_neighbours(self):
        """adj[f, b] = the face on the other side of the edge opposite to local vertex b of face f (-1 on a border edge).
        The reference intersects per-vertex face lists in Python for every affected Gaussian; here the edge -> faces
        relation is built once per mesh with a sort.  (On a non-manifold edge the reference takes `list(set)[0]`, i.e. an
        arbitrary member; this takes the smallest face id.)"""
        face = self.mesh.face
        key = (face.data_ptr(), face._version, tuple(face.shape))
        c = self.__dict__.get("_adj_cache")
        if c is not None and c[0] == key:
            return c[1]
        V, F = int(self.mesh.pos.shape[0]), int(face.shape[1])
        f = torch.arange(F, device=face.device)
        keys, owners = [], []
        for b in range(3):
            a, c2 = face[(b + 1) % 3], face[(b + 2) % 3]
            keys.append(torch.minimum(a, c2) * V + torch.maximum(a, c2))
            owners.append(f)
        keys, owners = torch.cat(keys), torch.cat(owners)                # entry e = b * F + f
        order = torch.argsort(keys * F + owners)                         # by edge, then by face id
        sk, so = keys[order], owners[order]
        start = torch.searchsorted(sk, keys)                             # first entry of each entry's edge group
        first = so[start]
        nxt = torch.clamp(start + 1, max=sk.numel() - 1)
        second = torch.where((start + 1 < sk.numel()) & (sk[nxt] == keys), so[nxt], torch.full_like(first, -1))
        adj = torch.where(first != owners, first, second).view(3, F).t().contiguous()   # [F, 3]
        self._adj_cache = (key, adj)
        return adj

    @torch.no_grad()
    def cleanup_barycentric_coordinates(self):
        """Re-assign every Gaussian whose barycentric coordinate went negative to the face across the offending edge, with
        distance-based coordinates there; on a border edge the coordinate is pushed back inside (the reference writes 0.005
        and then divides that scalar by its own sum, i.e. stores 1.0 -- reproduced).  Vectorised: the reference's per-Gaussian
        Python loop (one .item() per affected Gaussian, train_utils.py:306 every `bary_cleanup` iterations) becomes at
        most three batched passes (a row can have up to three negative coordinates; the loop handles them in column order
        with the ORIGINAL face of the row, which the passes replicate)."""
        mask = self.face_bary < 0
        if not bool(mask.any()):
            return
        adj = self._face_neighbours()
        orig_ids = self.face_ids.clone()
        xyz = self.get_xyz()
        rank = torch.cumsum(mask.to(torch.int64), dim=1) - 1
        for r in range(3):
            coord, bary = torch.where(mask & (rank == r))
            if coord.numel() == 0:
                break
            nf = adj[orig_ids[coord], bary]
            border = nf < 0
            if bool(border.any()):
                self.face_bary.data[coord[border], bary[border]] = 1.0
            if bool((~border).any()):
                cc, new_face = coord[~border], nf[~border]
                self.face_ids[cc] = new_face
                self.invalidate_caches()
                corners = self.mesh.pos[self.mesh.face[:, new_face].t()]            # [n, 3 (vertex), 3 (xyz)]
                dist = torch.linalg.norm(xyz[cc].unsqueeze(1) - corners, dim=2)
                self.face_bary.data[cc] = dist / dist.sum(dim=1, keepdim=True)

    # ---- gaussian_model.py:408-425 ------------------------------------------------------------------------------------
    def densify(self, max_grad, min_opacity, extent, max_screen_size):
        grads = self.pos_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)

    def prune(self, max_grad, min_opacity, extent, max_screen_size):
        mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            mask = torch.logical_or(torch.logical_or(mask, big_vs), big_ws)
        self.prune_points(mask)


def densification(gaussians, iteration, visibility_filter, radii, viewspace_point_tensor_grad, opt, cameras_extent):
    """train_utils.py:324-345: statistics every iteration, densify / prune on their intervals with the linearly annealed
    thresholds.  `opt` carries the OptimizationParams fields of arguments/__init__.py:109-150."""
    gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
    gaussians.add_densification_stats(viewspace_point_tensor_grad, visibility_filter)
    opacity_threshold = opt.opacity_threshold_fine_init - iteration * (
        opt.opacity_threshold_fine_init - opt.opacity_threshold_fine_after) / opt.densify_until_iter
    densify_threshold = opt.densify_grad_threshold_fine_init - iteration * (
        opt.densify_grad_threshold_fine_init - opt.densify_grad_threshold_after) / opt.densify_until_iter
    if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
        size_threshold = 20 if iteration > opt.opacity_reset_interval else None
        gaussians.densify(densify_threshold, opacity_threshold, cameras_extent, size_threshold)
    if iteration > opt.pruning_from_iter and iteration % opt.pruning_interval == 0:
        size_threshold = 20 if iteration > opt.opacity_reset_interval else None
        gaussians.prune(densify_threshold, opacity_threshold, cameras_extent, size_threshold)
