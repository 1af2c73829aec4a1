/*
 * include/csplat.h -- C-ABI of libcsplat.so, the MI355X (gfx950) hot path of cloth-splatting.
 *
 * Plain pointers and sizes only; every pointer named "device" is HBM memory owned by the caller
 * (torch tensors on the Python side), `stream` is a hipStream_t passed as void*.  All calls are
 * stream-ordered; the only host synchronisation is the 8-byte read of `num_rendered` (and the longest
 * tile list) inside csplat_forward (upstream reads num_rendered the same way): the scan kernel posts it
 * to a host-pinned mailbox that the host polls.  Return value: 0 = ok, non-zero = error
 * (text via csplat_last_error()); no C++ exception crosses this boundary.
 *
 * What each entry point replaces in the reference (/root/reference):
 *   csplat_forward / csplat_backward
 *       the `diff_gaussian_rasterization._C.rasterize_gaussians{,_backward}` calls behind
 *       GaussianRasterizer.forward, gaussian_renderer/__init__.py:16,76,156-164 (backward runs from
 *       loss.backward(), scene_reconstruction/train_utils.py:288).  Sources of that extension are an empty
 *       submodule (.gitmodules:7-9); the argument list mirrors its published rasterize_points.h.
 *   csplat_dist2
 *       `simple_knn._C.distCUDA2`, scene_reconstruction/gaussian_mesh.py:26,250; gaussian_model.py:20,134.
 *   csplat_gnn_*
 *       the torch_geometric MessagePassing gather / scatter-add inside InteractionNetwork.propagate,
 *       meshnet/graph_network.py:173-174 (gather x_i, x_j: PyG __lift__), :197 (concat), :136 (aggr='add').
 */
#ifndef CSPLAT_H
#define CSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSPLAT_ABI_VERSION 6   /* 6 (round 6): csplat_backward_views_parts / _slice_rows, csplat_gnn_edge_length_refine, csplat_rollout_head / _decode / _integrate, csplat_gnn_edge_features_ordered, csplat_gnn_rows_chain(_pack), csplat_binning_fields; round 4: csplat_view.busy_tiles / .valid, csplat_rows_dot_fwd's extra argument; 3: the binning chunk's layout (bbits, bmask); 4 (round 5): csplat_gather_words kind 2; 5: csplat_gnn_edge_mlp3* (e0_absmax, modes), csplat_absmax, csplat_linear_narrow128 */

/* scratch chunks requested through the allocator callback */
#define CSPLAT_CHUNK_GEOM 0    /* per-Gaussian state, kept for backward */
#define CSPLAT_CHUNK_BINNING 1 /* sorted tile instances, kept for backward */
#define CSPLAT_CHUNK_IMAGE 2   /* per-tile ranges, per-pixel n_contrib / final_T, kept for backward */
#define CSPLAT_CHUNK_TEMP 3    /* unsorted instances, sort ping-pong + histograms; may be freed after forward */
#define CSPLAT_CHUNK_TABLE 4   /* per-(workgroup, tile) counting table of the bucketed binning; may be freed after forward */

/* Must return a device pointer to at least `bytes` bytes, 256-byte aligned, valid until the matching
 * backward has run (GEOM/BINNING/IMAGE) or until csplat_forward returns (TEMP).  NULL = failure. */
typedef void *(*csplat_alloc_fn)(void *ctx, int chunk, size_t bytes);

int csplat_abi_version(void);
/* test hooks (results must not change beyond fp32 re-association).  bit 0: no culling -- every block mask of K5b is all ones;
 * bit 1: force the global radix-sort binning path instead of the tile-bucketed LDS sort;
 * bit 2: read R with a blocking stream synchronise instead of polling the pinned mailbox;
 * bit 3: unused (round 1's experimental four-wave forward was removed);
 * bit 4: quadruple the culling radius (sensitivity check of the culling bound);
 * bit 5: circle stage of the culling only (no exact ellipse-vs-box stage).  Bits 0 and 4 also switch the ellipse stage off;
 * bit 7: one K8 launch per view instead of one for all views of a step;
 * bit 8: bit-reproducible backward -- K7 stores one record per (list entry, quadrant) and every Gaussian sums its records in
 *        emission order instead of meeting in float atomics (csplat_backward_scratch_bytes grows accordingly: set the flag
 *        before sizing the scratch);
 * bit 9: per-view launches on per-view streams instead of one launch per stage for all views of a step;
 * bit 10: csplat_forward_views always waits for a call's counts before launching its second phase (no speculative launch with
 * the previous call's counts as capacities);
 * bit 11: tile sort by LSD radix only; bit 12: forced radix fallback of the bucket sort (both: test hooks of the sort's two paths);
 * bit 15: K6 in the row form (four survivors per step; default: sixteen per step) -- kept because its transmittance products run in list
 *         order, which the culling-exactness test holds bit for bit.
 * Bits 13, 14, 16-21 selected the shelved kernel forms of round 3 (K7 survivor columns, retire-waves flush, K6 items per wave / tile order,
 * block masks by blockIdx.y); those forms left the library in round 4 (history: commit 809fd4b) and the bits are
 * ignored.  The word is process-global and meant to be set between calls, not concurrently with them. */
int csplat_debug_flags(unsigned flags);
unsigned csplat_debug_flags_query(void);
/* measurement hook (not part of the operator interface): a device buffer the batched compositing backward fills with s_memtime stamps,
 * 12 uint64 per workgroup in launch order [view][workgroup] (tools/k7_stamps.py); NULL / 0 switches it off */
int csplat_debug_stamps(void *device_buffer, size_t bytes);
const char *csplat_last_error(void);

/* Sizes of the chunks (bytes) so that a caller may pre-allocate instead of answering the callback lazily. */
size_t csplat_geom_bytes(int P);
size_t csplat_image_bytes(int W, int H);
size_t csplat_binning_bytes(int64_t R, int W, int H);
size_t csplat_temp_bytes(int P, int64_t R, int W, int H);
size_t csplat_backward_scratch_bytes(int P, int64_t R); /* per-Gaussian accumulation records used by csplat_backward */

/* Byte offsets of the named sub-buffers inside a chunk (for tests / debugging; see DESIGN.md "HBM layout").
 * geom:    0 depth f32[P] | 1 xy f32[P][2] | 2 conic_opacity f32[P][4] | 3 rgb f32[P][3] | 4 cov3D f32[P][6]
 *          | 5 clamped u32[P] (bit c = channel c clamped) | 6 tiles_touched u32[P]
 *          | 7 offsets u32[P] (inclusive scan; filled only on the global radix-sort path)
 * binning: 0 keys u64[R] (sorted) | 1 ids u32[R] (sorted)   [ABI 3, behind them and not part of the offsets: the per-tile segment
 *          plan seg_offset[tiles+1] / slot_tile[slots]; the forward's checkpoints float4[slots][16 blocks][16 px] (T, colour so far
 *          at the start of every 256-entry list segment); the 16-bit block masks u16[R+1]; the tile-ordered records recA / recB
 *          float4[R+1], recC float2[R+1] (entry R = the null record); `bbits` u64[slots][16][4] = per (segment, 4x4 block) WHICH of the
 *          segment's entries the block blended (K6 writes, K7 reads); `bmask` u64[R/64+4][16] = the block masks transposed, per 64
 *          list entries and block (K5b writes, K6 reads).  R here is the LAYOUT count (csplat_view.layout_rendered >= num_rendered)]
 * image:   0 ranges i32[tiles][2] | 1 n_contrib u32[H*W] | 2 final_T f32[H*W]                               */
int csplat_geom_layout(int P, size_t *offsets8);
int csplat_binning_layout(int64_t R, int W, int H, size_t *offsets2);
/* byte offsets of ALL eleven sub-buffers of a BINNING chunk laid out for R list entries: 0 sorted keys, 1 sorted ids, 2 seg_offset[tiles + 1]
 * followed by blk_hi[tiles][16], 3 slot_tile, 4 checkpoints, 5 block masks, 6-8 the tile-ordered records, 9 bbits u64[slots][16][4] (which
 * entries of a 256-entry segment a 4x4 block blended), 10 bmask.  Diagnostics only (tools/k6_stats.py, bench.py's count of the (entry,
 * block) pairs = K7's float-atomic requests). */
int csplat_binning_fields(int64_t R, int W, int H, size_t *o11);
int csplat_image_layout(int W, int H, size_t *offsets3);

/* Forward: K1 preprocess (+ per-tile counts), K2 tile scan, K3 instance emission into tile buckets, K4 per-tile LDS sort
 * (global stable radix sort when a tile list exceeds 8192 entries), K5 segment plan, K6 compositing.
 *   means3D[P][3], shs[P][M][3] or NULL, colors_precomp[P][3] or NULL (exactly one of the two),
 *   opacities[P], scales[P][3]+rotations[P][4] or cov3D_precomp[P][6] (exactly one of the two),
 *   view/proj: the reference's transposed 4x4 matrices (flat index 4*row+col of world_view_transform /
 *   full_proj_transform, scene_reconstruction/cameras.py:63-67), campos[3], bg[3]  -- all device, fp32.
 *   D = active SH degree (0..3), M = SH coefficients per channel stored (16 for max degree 3).
 * Outputs (device): out_color[3][H][W], out_depth[1][H][W], radii[P] (int32).
 * *num_rendered (host) receives R = number of (Gaussian, tile) instances.
 * The three kept chunks are returned through geom/binning/image (device pointers from `alloc`). */
int csplat_forward(void *stream, int P, int D, int M, const float *bg, int W, int H, const float *means3D,
                   const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                   float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *view,
                   const float *proj, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                   csplat_alloc_fn alloc, void *alloc_ctx, float *out_color, float *out_depth, int32_t *radii,
                   int *num_rendered, void **geom, void **binning, void **image);

/* Two-phase form of csplat_forward for callers with several independent views in flight (new; upstream renders cameras one
 * by one: /root/reference/scene_reconstruction/train_utils.py:204-260).  _begin enqueues K1 and the counting half of the
 * binning on `stream` and returns without waiting; _finish (same thread or another) reads num_rendered -- the path's only
 * host round trip --, requests the R-sized chunks from the allocator given to _begin and enqueues K3..K6 on the same
 * stream.  Issue every _begin before the first _finish and the round trips and the compositing kernels of the views
 * overlap.  At most 64 tickets may be open; a ticket is released by _finish whatever it returns.
 * csplat_forward(...) == csplat_forward_begin(...) followed by csplat_forward_finish(...). */
int csplat_forward_begin(void *stream, int P, int D, int M, const float *bg, int W, int H, const float *means3D,
                         const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                         float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *view,
                         const float *proj, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                         csplat_alloc_fn alloc, void *alloc_ctx, int32_t *radii, int *ticket_out);
int csplat_forward_finish(int ticket, float *out_color, float *out_depth, int *num_rendered, void **geom,
                          void **binning, void **image);

/* Batched form: V independent views of one step in ONE call each way (the reference's loop over cameras,
 * /root/reference/scene_reconstruction/train_utils.py:204-260 forward, :288 backward).  Every view names its own stream;
 * the library fences them against `join_stream` on entry and exit (events), issues every view's K1/K2 before the first
 * num_rendered read, and in the backward runs the views' K7 concurrently.  accmask (CSPLAT_ACC_*): gradient outputs of
 * this view that are ADDED to a buffer an earlier view of the same call wrote (a parameter shared by several views then
 * needs no per-view temporaries and no summation launches); when any view accumulates, K8 of all views runs on the join
 * stream in view order.  dL_dmean2D / dL_dconic are always written.  Field meanings as in csplat_forward / csplat_backward. */
enum { CSPLAT_ACC_OPACITY = 1, CSPLAT_ACC_COLOR = 2, CSPLAT_ACC_MEAN3D = 4, CSPLAT_ACC_COV3D = 8, CSPLAT_ACC_SH = 16,
       CSPLAT_ACC_SCALE = 32, CSPLAT_ACC_ROT = 64,
       /* not an accumulate bit: the view's `scratch` records are ALL ZERO on entry and the call leaves them all zero again (K8 clears
        * every record it has consumed) -- a caller that keeps the buffer from step to step on one stream then needs no clearing launch per
        * step (25.6 MB of zero fill for four views of 100k Gaussians).  Without the bit the library clears the records itself. */
       CSPLAT_SCRATCH_ZEROED = 256 };
typedef struct csplat_view {
    void *stream;
    int P, D, M, W, H, prefiltered;
    float scale_modifier, tanfovx, tanfovy;
    const float *bg, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp, *view, *proj, *campos;
    void *alloc_ctx;
    float *out_color, *out_depth;   /* forward outputs (caller-allocated) */
    int32_t *radii;
    int num_rendered;               /* written by the forward: R, the number of (Gaussian, tile) instances */
    int layout_rendered;            /* written by the forward, read by the backward: the list capacity (>= R) the binning chunk
                                     * was laid out for -- csplat_forward_views may size it from the previous call's counts so
                                     * as not to wait for this call's (use it for csplat_backward_scratch_bytes too) */
    void *geom, *binning, *image;   /* chunk base pointers: written by the forward, read by the backward */
    const float *dL_dpix;           /* backward inputs */
    void *scratch;
    unsigned accmask;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
    int busy_tiles;                 /* written by the forward (0 = not known): tiles with a non-empty list -- the backward sizes the
                                     * compositing backward's grid with it (segments <= R / 256 + busy_tiles + 1) */
    const uint32_t *valid;          /* NULL, or (csplat_forward_views_faith) the device word that says whether the forward took place:
                                     * the backward of views[0..V) does nothing when it is 0 */
} csplat_view;
int csplat_forward_views(int V, csplat_view *views, csplat_alloc_fn alloc, void *join_stream);
/* The same call with its one host read (the views' counts) DEFERRED.  When the second phase can be launched on the previous call's
 * capacities (speculative launch, see above), csplat_forward_views_deferred returns right behind that launch with *pending = 1:
 * layout_rendered is filled in (the capacity), num_rendered is -1; the caller does the host work that does not need the counts while
 * the GPU runs K1..K6 and then calls csplat_forward_views_settle with the SAME array on the same join stream -- before
 * csplat_backward_views and before reading num_rendered.  settle: counts fit -> num_rendered filled in, *relaunched = 0; they do not ->
 * the second phase is repeated with exact sizes (new BINNING chunks through the allocator of the call, which must still be callable;
 * layout_rendered / binning updated) and *relaunched = 1.  *pending = 0: the call was complete, nothing to settle. */
int csplat_forward_views_deferred(int V, csplat_view *views, csplat_alloc_fn alloc, void *join_stream, int *pending);
int csplat_forward_views_settle(int V, csplat_view *views, void *join_stream, int *relaunched);
int csplat_backward_views(int V, csplat_view *views, void *join_stream);
/* csplat_backward_views cut into parts (round 6; no counterpart upstream -- the reference is single-GPU): parts bit 0 = the compositing
 * backward (K7) of all views, bit 1 = the per-Gaussian backward (K8) for slice `slice` of `nslices` equal ranges of Gaussians (boundaries
 * at multiples of 32; csplat_backward_slice_rows).  A view-parallel step launches K7 once, then the K8 slices one by one, and hands the
 * finished gradient rows of slice g to its collective while slice g + 1 computes.  Only the one-launch-per-stage path can be cut (views
 * sharing P, SH, scales and the image size); parts = 3, one slice = csplat_backward_views.  The parts together equal the whole call bit for
 * bit. */
int csplat_backward_views_parts(int n_views, csplat_view *views, void *join_stream, unsigned parts, int slice, int nslices);
/* rows [*row_lo, *row_hi) of the P Gaussians whose gradients K8 slice `slice` of `nslices` finishes */
int csplat_backward_slice_rows(int P, int slice, int nslices, int64_t *row_lo, int64_t *row_hi);
/* The batched forward WITHOUT a host read, for stream capture (a training step replayed as a hipGraph; the reference times its step
 * with an event pair around exactly such a loop body, train.py:146,178): both phases are launched with the caller's capacities --
 * caps[0] list entries per view, caps[1] longest tile list, caps[2] non-empty tiles, normally a previous call's counts plus a margin --
 * and *valid (device) becomes 1 when every view's counts fitted, else 0 (the second phase then left the views untouched, and
 * csplat_backward_views, which finds views[i].valid set, does nothing).  num_rendered is NOT the count (it is set to the capacity):
 * the counts stay on the device, at csplat_image_info_offset(W, H) inside each view's IMAGE chunk (u32 x 3).  The views must qualify
 * for the one-launch-per-stage path (2..8 views sharing P, SH, opacities, scales, image size); otherwise an error is returned. */
int csplat_forward_views_faith(int V, csplat_view *views, csplat_alloc_fn alloc, void *join_stream, const uint32_t *caps, uint32_t *valid);
size_t csplat_image_info_offset(int W, int H);

/* Backward: K7 compositing backward, K8 per-Gaussian backward.
 * out_color is the forward's colour image; dL_dpix[3][H][W] its gradient (the depth image carries no gradient,
 * as upstream).
 * scratch: device buffer of csplat_backward_scratch_bytes(P, R) bytes: one 64-byte-aligned accumulation record per Gaussian (9 floats
 * used).  ABI 3: K7 adds one 36-byte partial per (list entry, 4x4 pixel BLOCK that blended it) -- the nine lanes that hold the row sums
 * issue one float-atomic request to the Gaussian's record; K8 consumes the records.  In the bit-reproducible mode (csplat_debug_flags
 * bit 8) the buffer also holds one stored 9-float record per (list entry, block), summed per Gaussian in emission order.
 * Gradient outputs (device, fully overwritten): dL_dmean2D[P][3] (NDC units, .z = 0), dL_dconic[P][4],
 * dL_dopacity[P], dL_dcolor[P][3], dL_dmean3D[P][3], dL_dcov3D[P][6], dL_dsh[P][M][3] (may be NULL when
 * colors_precomp was used), dL_dscale[P][3], dL_drot[P][4] (may be NULL when cov3D_precomp was used). */
int csplat_backward(void *stream, int P, int D, int M, int R, const float *bg, int W, int H, const float *means3D,
                    const float *shs, const float *colors_precomp, const float *scales, float scale_modifier,
                    const float *rotations, const float *cov3D_precomp, const float *view, const float *proj,
                    const float *campos, float tanfovx, float tanfovy, const int32_t *radii, const void *geom,
                    const void *binning, const void *image, const float *out_color, const float *dL_dpix, void *scratch,
                    float *dL_dmean2D,
                    float *dL_dconic, float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D,
                    float *dL_dsh, float *dL_dscale, float *dL_drot);

/* distCUDA2: out[i] = mean of squared distances from point i to its 3 nearest other points. */
int csplat_dist2(void *stream, int P, const float *xyz, float *out);
/* the same result (bit for bit) in O(P * pruned candidates): Morton order + bounding-box pruning as the upstream extension
 * does; temp: csplat_dist2_temp_bytes(P) bytes of device memory. */
size_t csplat_dist2_temp_bytes(int P);
int csplat_dist2_ws(void *stream, int P, const float *xyz, float *out, void *temp);

/* Separable 11-tap window of the SSIM loss (utils/loss_utils.py:30-58), zero padded: out = G (x) G * in for every one of
 * the n_images [H][W] planes.  taps11 is a HOST pointer to the 11 normalised window weights.  Self-adjoint: the backward
 * of the operator is the operator.  (SURVEY.md 8(f) "next" row N2.) */
int csplat_blur11(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *in, float *out);

/* One Adam step (torch.optim.Adam semantics: no weight decay, no amsgrad, maximize = false) over n_tensors fp32 tensors in a
 * single launch per CSPLAT_ADAM_MAX_TENSORS tensors.  Host arrays of device pointers / element counts / per-tensor learning
 * rates (doubles, combined in double before the cast to fp32 as torch does; the reference keeps one parameter group per Gaussian attribute, /root/reference/scene_reconstruction/
 * gaussian_mesh.py:126-136, stepped at train_utils.py:310-319).  `step` is the 1-based step count AFTER this update. */
#define CSPLAT_ADAM_MAX_TENSORS 48
int csplat_adam_step(void *stream, int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                     float *const *exp_avg_sq, const int64_t *numel, const double *lr, double beta1, double beta2, double eps,
                     int64_t step);
/* The same step with NOTHING of the launch depending on a per-step host value (for a training step recorded into a hipGraph):
 * state_dev[0] (int32, device) = steps taken so far -- the kernel uses state_dev[0] + 1 for its bias corrections and a trailing
 * one-thread launch advances it; lr_dev = the tensors' learning rates as doubles in device memory; valid_dev (may be NULL) = a device
 * word: 0 there -> parameters, moments and the step count are left untouched.  At most CSPLAT_ADAM_MAX_TENSORS tensors. */
int csplat_adam_step_dev(void *stream, int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                         float *const *exp_avg_sq, const int64_t *numel, const double *lr_dev, double beta1, double beta2, double eps,
                         int *state_dev, const uint32_t *valid_dev);
/* dst[...] (float, device) = the concatenation of n <= 32 small device arrays, src[i] holding count[i] values of kind[i] (0 = float,
 * 1 = int32 / uint32 converted to float, exact below 2^24; 2 = int32 / uint32 copied BIT FOR BIT -- read the slot back as an integer:
 * instance counts exceed 2^24 on large scenes): one launch that collects a recorded step's log line -- step count, go / no-go
 * word, PSNR, loss, the views' instance counts -- for ONE copy to pinned host memory. */
int csplat_gather_words(void *stream, int n, const void *const *src, const int *kind, const int *count, float *dst);

/* Capacity-based densify / prune: replaces the boolean-mask indexing / torch.cat re-creation of every
 * parameter and Adam moment in /root/reference/scene_reconstruction/gaussian_model.py:266-341 and gaussian_mesh.py:336-431.
 *   csplat_mask_to_map: map[i] = base + (number of set mask bytes before i) where mask[i] != 0, else -1; *count_dev = number of
 *                       set bytes (device int32).  Stable.  temp: csplat_mask_to_map_temp_bytes(n) bytes.
 *   csplat_rows_scatter: for every tensor t and source row i with map[i] >= 0: dst[t][map[i]] = src[t][i] (row_bytes[t] bytes,
 *                       a multiple of 4); src[t] == NULL writes zeros (fresh Adam moments of appended rows).  ONE launch for up
 *                       to CSPLAT_ROWS_MAX_TENSORS tensors.  Source and destination rows must not overlap (ping-pong buffers
 *                       for compaction, rows beyond the live range for appends).  src / dst / row_bytes are HOST arrays. */
#define CSPLAT_ROWS_MAX_TENSORS 32
size_t csplat_mask_to_map_temp_bytes(int64_t n);
int csplat_mask_to_map(void *stream, int64_t n, const uint8_t *mask, int32_t base, int32_t *map, int32_t *count_dev, void *temp);
int csplat_rows_scatter(void *stream, int n_tensors, const void *const *src, void *const *dst, const int64_t *row_bytes,
                        int64_t n_rows, const int32_t *map);

/* Output layer of the time-conditioned simulator MLP, T time rows at once (/root/reference/meshnet/meshnet_network.py:
 * 339 `self.output = Linear(256, n_nodes * 3)`, applied at :367 to ONE time value per render() call; a training step makes
 * T = 3 such calls, train_utils.py:204-260):
 *   forward:  y[t][r] = b[r] + sum_k W[r][k] h[t][k]               W [R][K] row-major (torch Linear.weight), h [T][K], y [T][R]
 *   backward: dW[r][k] = sum_t dy[t][r] h[t][k], db[r] = sum_t dy[t][r], dh[t][k] = sum_r dy[t][r] W[r][k]   (deterministic)
 * K must be 256, 0 <= T <= 8.  scratch: csplat_rows_dot_scratch_bytes(T) bytes, not shared between concurrent calls. */
/* The two hidden layers in front of it (meshnet_network.py:337-338,364-366), for the same T time rows, one launch each way:
 *   forward : h1 = relu(e W1^T + b1) [T][256], h2 = relu(h1 W2^T + b2) [T][256]      e [T][K0] (the sinusoidal code, K0 <= 16),
 *             W1 [256][K0], W2 [256][256] row-major (torch Linear.weight)
 *   backward: from dh2 = dL/dh2 [T][256]: dW1 [256][K0], db1 [256], dW2 [256][256], db2 [256]   (e has no gradient: parameter-free code) */
int csplat_sim_hidden_fwd(void *stream, int T, int K0, const float *e, const float *W1, const float *b1, const float *W2, const float *b2,
                          float *h1, float *h2);
/* scratch: csplat_sim_hidden_scratch_bytes(T) bytes whose first word is zero on entry (the kernel leaves it zero), not shared between
 * calls that may run concurrently */
size_t csplat_sim_hidden_scratch_bytes(int T);
int csplat_sim_hidden_bwd(void *stream, int T, int K0, const float *e, const float *W2, const float *h1, const float *h2, const float *dh2,
                          float *dW1, float *db1, float *dW2, float *db2, void *scratch);
size_t csplat_rows_dot_scratch_bytes(int T);
/* add: NULL, or [T][R] added to the result (the simulator's `mesh_predictions[time_id] + residual`, meshnet_network.py:371) */
int csplat_rows_dot_fwd(void *stream, int T, int R, int K, const float *W, const float *b, const float *h, float *y, const float *add);
int csplat_rows_dot_bwd(void *stream, int T, int R, int K, const float *W, const float *h, const float *dy, float *dW, float *db,
                        float *dh, void *scratch);

/* The cloth regularisers of the reconstruction loss with their gradient in one pass (/root/reference/scene_reconstruction/
 * train_utils.py:83-102): D [T][V][3] deformed vertices of the step's T cameras, edge_index [2][E] int64, rest_len [E]:
 *   *loss = lambda_deform * 0.5 * (mean_v |D1-D0|_2 + mean_v |D2-D1|_2)          (only when T >= 3)
 *         + lambda_rigid * mean_{t,e} | rest_len[e] - |D[t][edge_index[1][e]] - D[t][edge_index[0][e]]|_2 |
 *         + lambda_momentum * mean_v |D2 - 2 D1 + D0|_1                          (only when T >= 3)
 *   grad [T][V][3] = d loss / d D (norms have gradient 0 at 0, as torch defines them).
 * scratch: csplat_cloth_regs_scratch_bytes(T, V, E) bytes, ZERO before the first call (every call leaves it reusable), not shared
 * between concurrent calls.  The loss value is summed in a fixed order.  Gradient: with the CSR of the graph (dst_rowptr / src_rowptr [V+1], dst_perm / src_perm [E], int32: edge ids
 * grouped by edge_index[1] / edge_index[0], ascending within a group) every vertex gathers its edges -- deterministic, no
 * atomics; with NULLs the edges scatter with float atomics (summation order not fixed). */
size_t csplat_cloth_regs_scratch_bytes(int T, int V, int64_t E);
int csplat_cloth_regs(void *stream, int T, int V, int64_t E, const float *D, const int64_t *edge_index, const float *rest_len,
                      float lambda_deform, float lambda_rigid, float lambda_momentum, float *loss, float *grad, void *scratch,
                      const int *dst_rowptr, const int *dst_perm, const int *src_rowptr, const int *src_perm);

/* Pixel coordinates of n world points (the `projections` by-product of render(), /root/reference/gaussian_renderer/__init__.py:
 * 166-179): hom = [p, 1] @ full_proj (device pointer to the row-major 4x4 the reference keeps as Camera.full_proj_transform),
 * ndc = hom.xy / hom.w, out = ((ndc + 1) * (W, H) - 1) / 2.  points [n][3], out_pixels [n][2]. */
int csplat_project_points(void *stream, int64_t n, const float *full_proj, int W, int H, const float *points, float *out_pixels);

/* psnr of /root/reference/utils/image_utils.py:19-21 per image: out[i] = 20 log10(1 / sqrt(mean((a_i - b_i)^2))) over the
 * n_per_image values (channels x pixels) of image i; a, b [n_images][n_per_image].  One launch; fixed summation order.
 * scratch: csplat_psnr_scratch_bytes(n_images) bytes, not shared between concurrent calls. */
size_t csplat_psnr_scratch_bytes(int64_t n_images);
int csplat_psnr(void *stream, int64_t n_images, int64_t n_per_image, const float *a, const float *b, void *scratch, float *out);

/* SSIM of /root/reference/utils/loss_utils.py:40-70 (window 11, sigma 1.5, zero padding, size_average) on n_images [H][W]
 * planes (n_images = batch * channels), fused:
 *   forward:  map[i] = SSIM(x, y)[i] (optional), partial[b] = sum of the map over workgroup b's tile
 *             (csplat_ssim_partial_count() floats; mean = sum(partial) / (n_images*H*W), summed by the caller in a fixed
 *             order), and p1, p2, p3 = dSSIM/dblur(x), dSSIM/dblur(x*x), dSSIM/dblur(x*y) per pixel (all three or NULL).
 *   backward: dx[i] = g_scalar[0] * inv_n * ( blur(p1) + 2 x blur(p2) + y blur(p3) )[i]   (gradient w.r.t. x only)
 *                     + add_scale[0] * addend[i]   when addend / add_scale (device pointers) are given: the gradient of the
 *             reference's whole image loss Ll1 + lambda_dssim (1 - ssim) (train_utils.py:50-74) leaves in one pass, addend =
 *             the sign(x - y) / n that csplat_l1 wrote.
 * taps11: the 11 host floats of the 1-D window, as csplat_blur11. */
size_t csplat_ssim_partial_count(int64_t n_images, int H, int W);
int csplat_ssim_fwd(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *x, const float *y,
                    float *p1, float *p2, float *p3, float *map_out, float *partial);
int csplat_ssim_bwd(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *x, const float *y,
                    const float *p1, const float *p2, const float *p3, const float *g_scalar, float inv_n,
                    const float *addend, const float *add_scale, float *dx);
/* masked form of the reference's image loss (train_utils.py:61-67: `((1.0 - ssim_map) * mask_tensor).mean()`): x, y
 * [n_batch][channels][H][W], mask [n_batch][mask_channels][H][W] with mask_channels = 1 (Camera.mask, broadcast over the
 * colour channels) or = channels.  partial sums hold sum((1 - SSIM) * mask) over the workgroup's tile (the loss term is
 * sum(partial) / (n_batch*channels*H*W)); p1..p3 are pre-multiplied by the mask, so csplat_ssim_bwd (g_scalar = MINUS the
 * incoming gradient of the loss term) serves both forms. */
int csplat_ssim_fwd_masked(void *stream, int64_t n_batch, int channels, int H, int W, const float *taps11, const float *x,
                           const float *y, const float *mask, int mask_channels, float *p1, float *p2, float *p3,
                           float *map_out, float *partial);
/* The densification statistics of a training step (/root/reference/scene_reconstruction/train_utils.py:276-285:
 * `radii = torch.cat(radii_list, 0).max(dim=0).values`, `visibility_filter = torch.cat(visibility_filter_list).any(dim=0)`, the
 * viewspace-gradient sum over the step's cameras) in one launch.  mean2d_grads / radii: HOST arrays of V (<= 16) device pointers,
 * [P][3] float / [P] int32 each (a NULL gradient entry counts as zero); outputs: grad_sum [P][3], radii_max [P], visible [P] (0 / 1
 * bytes).  grad_sum or (radii_max, visible) may be NULL. */
int csplat_step_stats(void *stream, int64_t P, int V, const float *const *mean2d_grads, const int *const *radii, float *grad_sum,
                      int *radii_max, uint8_t *visible);
/* The whole image loss of a training step in two launches forward, one backward -- the reference's Ll1 + lambda_dssim * (1 - ssim) (masked:
 * mean|(x - y) m| + lambda_dssim * mean((1 - ssim_map) m)), /root/reference/scene_reconstruction/train_utils.py:50-74, the per-camera
 * PSNR it logs every step (:262-283, utils/image_utils.py:17-21, unmasked) and the sum with one more device scalar (the regularisers,
 * :76-237):
 *   out[0] = img_weight * image_loss + add_weight * add[0]   (add may be NULL)      out[1] = psnr_scale * sum_b PSNR_b
 *   out[2] = image_loss                                                              out[3] = Ll1
 * x, y [n_batch][channels][H][W]; mask NULL or [n_batch][mask_channels][H][W] (mask_channels 1 or channels).  p1..p3 (float images) and
 * sign8 (one byte per element) are what csplat_image_loss_bwd needs; all four NULL = no gradient wanted.  scratch:
 * csplat_image_loss_scratch_bytes bytes (no initial state, not shared between concurrent calls).  Bit-reproducible (workgroup
 * partials summed in index order by a one-workgroup second launch).
 * csplat_image_loss_bwd: dx = g_scalar[0] * img_weight * d(image_loss)/dx. */
size_t csplat_image_loss_scratch_bytes(int64_t n_batch, int channels, int H, int W);
int csplat_image_loss_fwd(void *stream, int64_t n_batch, int channels, int H, int W, const float *taps11, const float *x,
                          const float *y, const float *mask, int mask_channels, float lam, float img_weight, const float *add,
                          float add_weight, float psnr_scale, float *p1, float *p2, float *p3, signed char *sign8, void *scratch,
                          float *out);
int csplat_image_loss_bwd(void *stream, int64_t n_batch, int channels, int H, int W, const float *taps11, const float *x,
                          const float *y, const float *p1, const float *p2, const float *p3, const signed char *sign8,
                          const float *mask, int mask_channels, float lam, float img_weight, const float *g_scalar, float *dx);
/* l1_loss of /root/reference/utils/loss_utils.py:20-23 with its gradient in the same pass:
 *   *loss = mean_i |a[i] - b[i]|,   grad[i] = sign(a[i] - b[i]) / n   (grad may be NULL).
 * csplat_l1_masked: *loss = mean_i |(a[i] - b[i]) * m[i]|, grad[i] = sign((a-b)*m) * m / n, with the mask laid out as for
 * csplat_ssim_fwd_masked (hw = H*W values per plane).
 * scratch: csplat_l1_scratch_bytes() bytes whose last word is zero on entry (the kernel restores it), not shared between
 * calls that may run concurrently.  Deterministic (fixed summation order). */
size_t csplat_l1_scratch_bytes(void);
int csplat_l1(void *stream, int64_t n, const float *a, const float *b, void *scratch, float *loss, float *grad);
int csplat_l1_masked(void *stream, int64_t n_batch, int channels, int64_t hw, const float *a, const float *b, const float *mask,
                     int mask_channels, void *scratch, float *loss, float *grad);
/* The same loss for a caller that runs the backward later (autograd, `loss.backward()` of train_utils.py:288): the forward keeps ONE
 * BYTE per element, sign8[i] = sign((a[i] - b[i]) * m[i]) in {-1, 0, 1}, instead of a float gradient image, and
 *   csplat_l1_signs_bwd: out[i] = g_scalar[0] * sign8[i] * m[i] / n     (g_scalar: the incoming gradient of the loss, a device scalar)
 * is the whole backward -- no separate multiply by the incoming gradient.  mask may be NULL (then mask_channels is ignored). */
int csplat_l1_signs(void *stream, int64_t n_batch, int channels, int64_t hw, const float *a, const float *b, const float *mask,
                    int mask_channels, void *scratch, float *loss, signed char *sign8);
int csplat_l1_signs_bwd(void *stream, int64_t n_batch, int channels, int64_t hw, const signed char *sign8, const float *mask,
                        int mask_channels, const float *g_scalar, float *out);

/* Fused mesh -> Gaussian transform (SURVEY.md 8(f) "next" row N1): MultiGaussianMesh.get_xyz + get_rotation,
 * scene_reconstruction/gaussian_mesh.py:151-188.  face_vertex_ids[P][3] (int64, device) = mesh.face[:, face_ids].T.
 *   csplat_mesh_rest       per-Gaussian constants of the REST face (in-plane basis, normal, in-plane coordinates) into
 *                          `rest` (csplat_mesh_rest_bytes(P) bytes); recompute when the mesh / face assignment changes
 *   csplat_mesh_transform_fwd   out_xyz[P][3] = barycentric centre on the deformed vertices; out_quat[P][4] =
 *                          normalize(rotation) (x) quaternion(Kabsch(rest face -> deformed face)), roma XYZW convention
 *   csplat_mesh_transform_bwd   gradients w.r.t. vertices[V][3] (atomically scattered, zeroed here), bary[P][3], rotation[P][4] */
size_t csplat_mesh_rest_bytes(int P);
int csplat_mesh_rest(void *stream, int P, const int64_t *face_vertex_ids, const float *rest_vertices, void *rest);
int csplat_mesh_transform_fwd(void *stream, int P, const int64_t *face_vertex_ids, const float *vertices, const float *bary,
                              const float *rotation, const void *rest, float *out_xyz, float *out_quat);
int csplat_mesh_transform_bwd(void *stream, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                              const float *bary, const float *rotation, const void *rest, const float *g_xyz,
                              const float *g_quat, float *d_vertices, float *d_bary, float *d_rotation);
/* The same for the T cameras of a training step in one launch each way (the reference calls get_xyz / get_rotation once per
 * render(), train_utils.py:204-260): vertices [T][V][3] -> out_xyz [T][P][3], out_quat [T][P][4]; backward takes g_xyz
 * [T][P][3] / g_quat [T][P][4] (either may be NULL) and returns d_vertices [T][V][3] and d_bary / d_rotation summed over the
 * cameras in camera order.  Vertex gradients: with vertex_rowptr [V+1] / vertex_corners [3P] (int32; the (Gaussian, corner)
 * pairs incident to each vertex as 3 * gaussian + corner, grouped by vertex: a stable sort of face_vertex_ids) and
 * corner_scratch (T*P*9 floats) they are gathered in that fixed order -- deterministic, no atomics; with NULLs they are
 * scattered with float atomics. */
int csplat_mesh_transform_fwd_views(void *stream, int T, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                                    const float *bary, const float *rotation, const void *rest, float *out_xyz, float *out_quat);
int csplat_mesh_transform_bwd_views(void *stream, int T, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                                    const float *bary, const float *rotation, const void *rest, const float *g_xyz,
                                    const float *g_quat, float *d_vertices, float *d_bary, float *d_rotation,
                                    const int *vertex_rowptr, const int *vertex_corners, float *corner_scratch);

/* ---- in-library kernel timing (HIP events on the launch stream; used by bench.py's roofline leg) ------
 * mask bit k enables bracketing of kernel class k with a start/stop event pair on the stream it is launched on:
 *   0 K1 preprocess | 1 K2 scan | 2 K3 key emission | 3 K4 radix sort (all passes) | 4 K5 tile ranges
 *   5 K6 compositing fwd | 6 K7 compositing bwd | 7 K8 preprocess bwd | 8 distCUDA2 | 9 GNN kernels
 * csplat_prof_read synchronises the recorded events of class k, returns their summed duration (ms) and the
 * number of brackets, and recycles the events. */
int csplat_prof_enable(unsigned mask);
int csplat_prof_read(int kernel_class, double *ms_total, int64_t *launches);

/* ---- MeshNet message passing (meshnet/graph_network.py:151-222) ------------------------------------
 * Graph structure is passed as two CSR orderings of the E directed edges, built once per graph by
 * csplat_gnn_build_csr: for key in {dst = edge_index[1], src = edge_index[0]}:
 *   rowptr[N+1] (int32) and perm[E] (int32) listing edge ids grouped by key, ascending edge id inside a row
 *   (=> a fixed, reproducible summation order; no float atomics on this path).
 * Rows of L floats move as float4 (16 B per lane) when L % 4 == 0, as scalars otherwise. */
size_t csplat_gnn_csr_temp_bytes(int N, int64_t E);
int csplat_gnn_build_csr(void *stream, int N, int64_t E, const int64_t *keys /* device, [E] */, int32_t *rowptr,
                         int32_t *perm, void *temp);

/* edge pre-activation of the first edge-MLP layer with the concat folded away:
 *   out[e][:] = relu?( xa[dst[e]][:] + xb[src[e]][:] + ec[e][:] )          (ec already holds e@W_e^T + b)
 * xa = x @ W_i^T, xb = x @ W_j^T are node-level products (N x L); may run in place (out == ec). */
int csplat_gnn_edge_combine_fwd(void *stream, int N, int64_t E, int L, const int64_t *edge_index /* [2][E] */,
                                const float *xa, const float *xb, const float *ec, int relu, float *out);
/* backward of the above w.r.t. xa, xb (gradient w.r.t. ec is g itself, masked by relu):
 *   g_masked = g * (out > 0) if relu;  dxa[n] = sum_{e: dst(e)=n} g_masked[e];  dxb[n] = sum_{e: src(e)=n} g_masked[e] */
int csplat_gnn_edge_combine_bwd(void *stream, int N, int64_t E, int L, const float *g, const float *out, int relu,
                                const int32_t *rowptr_dst, const int32_t *perm_dst, const int32_t *rowptr_src,
                                const int32_t *perm_src, float *g_masked, float *dxa, float *dxb);
/* aggr='add': agg[n][:] = sum_{e in row n} msg[perm[e]][:]   (deterministic segmented sum) */
int csplat_gnn_segment_sum(void *stream, int N, int64_t E, int L, const float *msg, const int32_t *rowptr,
                           const int32_t *perm, float *agg);
/* its backward: dmsg[e][:] = dagg[key[e]][:]   (row gather) */
int csplat_gnn_gather_rows(void *stream, int64_t E, int L, const float *rows, const int64_t *keys, float *out);
/* the same pass also leaves *absmax = max |value| over the gathered rows (L a multiple of 4; what csplat_gnn_edge_mlp3's mode 0 scales by) */
int csplat_gnn_gather_rows_absmax(void *stream, int64_t E, int L, const float *rows, const int64_t *keys, float *out, float *absmax);
/* Edge features of one rollout step (PyG Cartesian(norm=False) + Distance(norm=False), the `transformer(graph)` of
 * /root/reference/train_meshnet_sim.py:152 and dataloader_sim.py): out[e] = (pos[row] - pos[col], |pos[row] - pos[col]|) with
 * row = edge_index[0][e], col = edge_index[1][e]; pos [N][3], out [E][4] (16-byte aligned). */
int csplat_gnn_edge_features(void *stream, int64_t E, const float *pos, const int64_t *edge_index, float *out);
/* A chain of 128-wide Linears on node rows with pre-packed 16-bit-piece weights (the node update's machinery; round 6):
 *   mode 0: out_a = x Wa^T, out_b = x Wb^T           -- the first processor layer's x_i / x_j products (graph_network.py:178-199)
 *   mode 1: out_a = relu(W1 relu(W0 x + b0) + b1)    -- the decoder's two hidden layers (graph_network.py:295-332); out_b unused
 * x, out_* [N][128] fp32, 16-byte aligned, outputs not aliasing x; image = csplat_gnn_rows_chain_pack(mode, first, second) of
 * csplat_gnn_node_update_image_bytes() bytes, packed and used under one csplat_gnn_edge_mlp3_mode. */
int csplat_gnn_rows_chain_pack(void *stream, int mode, const float *Wfirst, const float *Wsecond, void *image);
int csplat_gnn_rows_chain(void *stream, int64_t N, int mode, const float *x, const void *image, const float *b0, const float *b1,
                          float *out_a, float *out_b);
/* csplat_gnn_edge_features for the edges in another order: row r of out = the features of edge order[r] (the rollout encodes its edges in the
 * destination order of GraphCSR.agg_plan: a permuted read of the [E] index pairs instead of a gather of [E][4] rows afterwards). */
int csplat_gnn_edge_features_ordered(void *stream, int64_t E, const float *pos, const int64_t *edge_index, const int64_t *order, float *out,
                                     float *absmax /* or NULL: receives max |value| (as csplat_absmax: atomicMax on the bits; zeroed by the caller) */);
/* Head and tail of ONE rollout step of ClothMeshSimulator around the network's launches (round 6: a recorded step holds library kernels only).
 *   head:      feats[n] = normalise(cat(hist[0][n], .., hist[H-1][n], one_hot(node_type[n], T)))   (/root/reference/meshnet/cloth_network.py:72-110;
 *              mean / std [3H + T] or both NULL = IdentityNormalizer); also *counter += 1 (the step's number + 1, device side)
 *   decode:    v[n] = last_v[n] + denormalise(W h[n] + b)   (the decoder's last Linear 128 -> D <= 4 and cloth_network.py:163-193); *fine = 0
 *              when a row is not finite (the fp16-piece arithmetic's overflow signal, meshnet/graph_network.py)
 *   integrate: v[grasped] = actions[*counter - 1]; preds[*counter - 1] = v; pos += v; hist <- (hist[1:], v)
 *              (/root/reference/train_meshnet_sim.py:176,256-262).  hist [H][N][D], pos / v [N][D], actions / preds [steps][..]. */
int csplat_rollout_head(void *stream, int N, int H, int T, const float *hist, const int32_t *node_type, const float *mean, const float *stdv,
                        float *feats, int32_t *counter, float *absmax /* or NULL: max |feature|, as csplat_absmax */);
int csplat_rollout_decode(void *stream, int N, int D, const float *h, const float *W, const float *b, const float *omean, const float *ostd,
                          const float *last_v, float *v, int32_t *fine);
int csplat_rollout_integrate(void *stream, int N, int H, int D, float *v, const float *actions, const int32_t *counter, int64_t grasped,
                             float *pos, float *hist, float *preds, float *absmax2 /* or NULL: two words set to zero for the next step's head / edge features */);
/* The `real_world` branch of the rollout (/root/reference/train_meshnet_sim.py:211-250: per rollout step, ten iterations of a fresh
 * torch.optim.Adam(lr = 1e-3) on the predicted velocities against sum_e w_e (|(pos + v)[row_e] - (pos + v)[col_e]| - rest_len_e)^2).
 * v [N][3] is updated in place; edge_w [E] or NULL (the reference zeroes ONE deviation: `length_deviation[grasped_particle] *= 0`);
 * the CSR orderings by destination (edge_index[1]) and by source (edge_index[0]) as csplat_gnn_build_csr leaves them; scratch = 9 N
 * floats.  One launch per iteration (gradient gathered per node over both orderings -- no atomics -- and the Adam step of the node's own
 * coordinates in the same pass), nothing read back. */
int csplat_gnn_edge_length_refine(void *stream, int N, int64_t E, const float *pos, float *v, const int64_t *edge_index, const float *rest_len,
                                  const float *edge_w, const int32_t *dst_rowptr, const int32_t *dst_perm, const int32_t *src_rowptr,
                                  const int32_t *src_perm, int iters, double lr, double beta1, double beta2, double eps, float *scratch);

/* The 128-wide Linear layers of the MeshNet MLPs for inference (replaces the cuBLAS sgemm behind nn.Linear at
 * /root/reference/meshnet/graph_network.py:198,221 when no autograd graph is recorded), with everything that follows the
 * product fused into the accumulators:
 *   out[m][:] = LN?( relu?( alpha * (A[m][:] @ W^T) + bias + gather_a[index_a[m]][:] + gather_b[index_b[m]][:]
 *                           + add_pre[m][:] ) ) + add_post[m][:]
 * A [M][128], W [128][128] row-major (torch Linear.weight), bias [128] or NULL, gather_* [*][128] with int64 row indices
 * (both or neither), ln_gamma / ln_beta [128] (both or neither; biased variance, ln_eps), add_pre / add_post [M][128] or
 * NULL (not together with the gathers), out [M][128]; out may alias A (and add_post may then alias both).
 * fp32 throughout (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate). A, W and out 16-byte aligned. */
/* The node update of one InteractionNetwork layer (/root/reference/meshnet/graph_network.py:203-222, two hidden layers of
 * width 128) in one launch, optionally followed by the next layer's node-level products:
 *   h = relu(agg @ Wa^T + x @ Wx^T + b0);  h = relu(h @ W2^T + b2);  x_new = LayerNorm(h @ W3^T + b3) + x;
 *   xa_next = x_new @ Wi_next^T;  xb_next = x_new @ Wj_next^T          (Wi_next, Wj_next both NULL: skipped)
 * agg, x, x_new, xa_next, xb_next [N][128]; all weights [128][128] row-major, contiguous (Wa / Wx are the two column
 * blocks of the first Linear's weight, cut by the caller); exact fp32 MFMA.  x_new must not alias agg or x. */
int csplat_gnn_node_update(void *stream, int64_t N, const float *agg, const float *x, const float *Wa, const float *Wx,
                           const float *b0, const float *W2, const float *b2, const float *W3, const float *b3,
                           const float *ln_gamma, const float *ln_beta, float ln_eps, const float *Wi_next,
                           const float *Wj_next, float *x_new, float *xa_next, float *xb_next);

/* The same node update on 16-bit pieces -- by csplat_gnn_edge_mlp3_mode: 0 (default) two fp16 pieces per operand, three products, everything run
 * at a fixed 2^-4 scale (values up to ~1e6 in magnitude fit; 1e-5 of the output scale against fp64 held from 1e-2 to 300 times the rollout's
 * magnitudes in tests/test_knn_gnn_gpu.py); 1 three bf16 pieces, six products (fp32's exponent range) -- with the six weight matrices
 * PRE-PACKED as MFMA A operands (pack under the mode the launch will run in): csplat_gnn_node_update_pack lays Wa, Wx, W2, W3 and the
 * next layer's Wi, Wj (both NULL: none; all [128][128] row-major, contiguous) out in `image` (csplat_gnn_node_update_image_bytes() bytes
 * of device memory, 16-byte aligned) -- once per weight version; csplat_gnn_node_update_packed is the launch (has_next = the image holds
 * Wi / Wj and xa_next / xb_next are written).  23 us (mode 0) / 30 us (mode 1) against 44 for csplat_gnn_node_update at N = 1e4.
 * piece_ptr != NULL: `agg` holds the PIECES csplat_gnn_edge_mlp3's fused aggregation left, node v's aggregate = the sum of pieces
 * piece_ptr[v] .. piece_ptr[v + 1] - 1 in that order (int32 [N + 1]) -- formed while the rows are loaded, no csplat_gnn_segment_sum launch. */
size_t csplat_gnn_node_update_image_bytes(void);
int csplat_gnn_node_update_pack(void *stream, const float *Wa, const float *Wx, const float *W2, const float *W3, const float *Wi_next,
                                const float *Wj_next, void *image);
int csplat_gnn_node_update_packed(void *stream, int64_t N, const float *agg, const float *x, const void *image, const float *b0,
                                  const float *b2, const float *b3, const float *ln_gamma, const float *ln_beta, float ln_eps,
                                  int has_next, float *x_new, float *xa_next, float *xb_next, const int32_t *piece_ptr);

/* LayerNorm(128) of the MeshNet MLPs under autograd (/root/reference/meshnet/graph_network.py:86-97,139-150: every edge / node
 * MLP ends in nn.LayerNorm), single HBM passes over [M][128] rows:
 *   csplat_ln128_fwd   y = (x - mean) * rstd * gamma + beta; stats[row] = (mean, rstd)  (biased variance, as torch)
 *   csplat_ln128_bwd   dx; dgamma[128] = sum_rows g * xhat; dbeta[128] = sum_rows g  (fixed summation order: deterministic);
 *                      dxsum[128] (or NULL) = column sums of dx: the bias gradient of the Linear layer whose output was normalised;
 *                      g_rows[M] (or NULL): row r of the incoming gradient is g[g_rows[r]] -- the backward of the segmented sum
 *                      that follows the edge LayerNorm (every edge reads its destination node's row) without a gathered copy;
 *                      x_normalized != 0: `x` holds xhat = (x - mean) * rstd itself (what csplat_linear128_ex's LayerNorm
 *                      epilogue leaves with gamma = 1, beta = 0), stats only supplies rstd
 *   csplat_relu_mask_bias128   gm = out > 0 ? g : 0 and dbias[128] = column sums of gm: ReLU backward + bias gradient of a
 *                      Linear + ReLU layer in one pass (out NULL: no mask; gm NULL: sums only)
 * partials: 3 x csplat_ln128_partial_floats(M) floats of scratch (1 x for csplat_relu_mask_bias128). */
size_t csplat_ln128_partial_floats(int64_t M);
int csplat_ln128_fwd(void *stream, int64_t M, const float *x, const float *gamma, const float *beta, float eps, float *y, float *stats);
int csplat_ln128_bwd(void *stream, int64_t M, const float *g, const float *x, const float *stats, const float *gamma, float *dx,
                     float *dgamma, float *dbeta, float *dxsum, const int64_t *g_rows, int x_normalized, float *partials);
int csplat_relu_mask_bias128(void *stream, int64_t M, const float *g, const float *out, float *gm, float *dbias, float *partials);

/* Weight gradient of a 128 -> 128 Linear layer under autograd: dW[o][i] = sum_e g[e][o] * x[e][i], g and x [M][128] row-major,
 * dW [128][128] (torch Linear.weight layout).  Exact-fp32 MFMA, split over row slices, partial results summed in slice order
 * (deterministic).  workspace: csplat_dw128_workspace_bytes(M) bytes. */
size_t csplat_dw128_workspace_bytes(int64_t M);
int csplat_dw128(void *stream, int64_t M, const float *g, const float *x, float *dW, void *workspace);
/* the same with dbias[128] = column sums of g (the bias gradient, from the g rows the product reads anyway) and, when x_relu != 0,
 * max(x, 0) in place of x (a layer whose saved input is the PRE-activation of the ReLU in front of it) */
int csplat_dw128_bias(void *stream, int64_t M, const float *g, const float *x, int x_relu, float *dW, float *dbias, void *workspace);

/* how csplat_linear128 forms its products: 0 = v_mfma_f32_32x32x2_f32 (exact fp32 products); 1 = three bf16 pieces per
 * operand and the six significant partial products on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; error ~3 x 2^-24 relative
 * per term -- measured rms error vs fp64 1.2e-7, the fp32-MFMA path and the library sgemm 1.5e-7 --; 6/16 of the
 * matrix-core time).  Default 1.  Applies to the persistent (edge-level) kernel. */
int csplat_linear128_mode(unsigned mode);
unsigned csplat_linear128_mode_query(void);   /* the mode currently in force */
int csplat_linear128(void *stream, int64_t M, const float *A, const float *W, const float *bias, float alpha, int relu,
                     const float *gather_a, const int64_t *index_a, const float *gather_b, const int64_t *index_b,
                     const float *ln_gamma, const float *ln_beta, float ln_eps, const float *add_pre,
                     const float *add_post, float *out);
/* The same product for the autograd path, where the weight is seldom a contiguous [128][128] matrix of its own: W is read as
 * W[j * ldw + k] (w_transposed = 0: a Linear.weight, or a 128-column slice of a wider one, ldw = its row stride) or as W[k * ldw + j]
 * (w_transposed = 1: the transpose, i.e. the input-gradient product g @ W, without materialising W^T); and `mask` [M][128] (or NULL)
 * zeroes the outputs whose mask entry is not positive, after everything else: the ReLU backward of the layer whose saved
 * output is handed in, folded into the GEMM that produces its incoming gradient.  ln_stats [M][2] (or NULL): with a LayerNorm
 * epilogue, the (mean, rstd) of every row -- what csplat_ln128_bwd needs, so that training can take the fused epilogue too. */
int csplat_linear128_ex(void *stream, int64_t M, const float *A, const float *W, int ldw, int w_transposed, const float *bias,
                        float alpha, int relu, const float *gather_a, const int64_t *index_a, const float *gather_b,
                        const int64_t *index_b, const float *ln_gamma, const float *ln_beta, float ln_eps,
                        const float *add_pre, const float *add_post, const float *mask, float *ln_stats, float *out);

/* out[M,128] = (ReLU)(x[M,K] W^T + bias) for a narrow input, 1 <= K <= 32: the first Linear of the encoders' MLPs
 * (/root/reference/meshnet/graph_network.py:48-111, build_mlp's NN-0 on 4 edge / 8 node features).  W is [128][ldw >= K] row-major as
 * nn.Linear stores it, x is [M][ldx >= K]; bias may be NULL; out is dense [M][128], 16-byte aligned. */
int csplat_linear_narrow128(void *stream, int64_t M, int K, const float *x, int ldx, const float *W, int ldw, const float *bias, int relu,
                            float *out);

/* The WHOLE edge MLP of an InteractionNetwork layer in one launch (inference; csplat_edge_mlp.hip) -- replaces the three csplat_linear128
 * calls of rounds 1-4 for /root/reference/meshnet/graph_network.py:178-199 (`message`: LN(MLP(cat[x_i, x_j, e]))):
 *     out[e] = LayerNorm( W2 relu( W1 relu( alpha * W0 e0[e] + b0 + xa[index_a[e]] + xb[index_b[e]] ) + b1 ) + b2 ) * gamma + beta
 * e0 / out [E][128] fp32 (out != e0), xa / xb [N][128] (the node-level x_i / x_j column-block products of the first Linear; N < 2^23),
 * index_a / index_b int64 [E], alpha a power of two (the edge scale 2^l of SURVEY F7).  `image` = the three weight matrices as
 * csplat_gnn_edge_mlp3_pack lays them out (csplat_gnn_edge_mlp3_image_bytes() bytes of device memory, 16-byte aligned): per wave of the
 * kernel the 16-bit pieces of its 32 output rows of the three W as MFMA A operands, in the order the kernel loads them into registers.
 * Pack once per weight version AND mode: W_l is read as W_l[j * ld_l + k] (a Linear.weight or a 128-column slice of a wider one).
 * csplat_gnn_edge_mlp3_mode(mode) selects the arithmetic and returns the previous mode (any other value only queries):
 *   0 (default)  two fp16 pieces per operand, three products: 3e-7 of the output scale against fp64 inside its domain -- the values are
 *                brought into fp16's range by a power of two taken from e0_absmax = the device word csplat_absmax leaves (max |e0| over
 *                the launch's rows or over a superset of them; NULL = 1.0).  Domain and failure mode: header of csplat_edge_mlp.hip.
 *   1            three bf16 pieces per operand, six products: fp32's exponent range, 7e-7 against fp64; e0_absmax is not read.
 * csplat_absmax(n, x, out): *out = max |x[i]| (n a multiple of 4, x 16-byte aligned; one pass, asynchronous on `stream`).
 * Fused aggregation (mode 0; pieces != NULL, out may be NULL and is not written): the rows are in DESTINATION order (index_a
 * non-decreasing: the caller permuted e0 / index_a / index_b by the CSR order), and instead of E message rows the launch writes their sums
 * over runs of equal index_a, cut additionally every 8 rows: piece p = sum of the rows of run p, [npieces][128] fp32, numbered in row
 * order; group_piece0[g] = number of the first piece of rows 8g .. 8g + 7 (int32 [ceil(E / 8)]).  A node's aggregate (graph_network.py:
 * 201-222, aggr = 'add') is the sum of its consecutive pieces -- csplat_gnn_segment_sum over the piece rows; one writer per piece, fixed
 * order of addition: deterministic.  Replaces the E x 128 write here and the E x 128 read of the segmented sum by ~(E / 8 + N) x 128. */
int csplat_gnn_edge_mlp3_mode(int mode);
size_t csplat_gnn_edge_mlp3_image_bytes(void);
int csplat_gnn_edge_mlp3_pack(void *stream, const float *W0, int ld0, const float *W1, int ld1, const float *W2, int ld2, void *image);
int csplat_absmax(void *stream, int64_t n, const float *x, float *out);
int csplat_gnn_edge_mlp3(void *stream, int64_t E, const float *e0, float alpha, const float *e0_absmax, const float *xa,
                         const int64_t *index_a, const float *xb, const int64_t *index_b, const void *image, const float *b0,
                         const float *b1, const float *b2, const float *ln_gamma, const float *ln_beta, float ln_eps, float *out,
                         const int32_t *group_piece0, float *pieces);

/* The same kernel as a stand-alone operator on NARROW input rows -- the encoders' MLPs (/root/reference/meshnet/graph_network.py:48-111:
 * build_mlp(K -> 128 -> 128 -> 128) + LayerNorm) in ONE launch:
 *     out[m] = LayerNorm( W2 relu( W1 relu( W0 x[m] + b0 ) + b1 ) + b2 ) * gamma + beta
 * x [M][K] fp32 (K a multiple of 4, 4 .. 128; M <= 2^22), out [M][128].  `image` = csplat_gnn_edge_mlp3_pack(W0 PADDED with zero columns
 * to [128][128], W1, W2) under mode 0; x_absmax = csplat_absmax of x (or of a superset; NULL = 1.0).  Mode 0 (fp16 pieces) only. */
int csplat_gnn_mlp3_rows(void *stream, int64_t M, const float *x, int K, const float *x_absmax, const void *image, const float *b0,
                         const float *b1, const float *b2, const float *ln_gamma, const float *ln_beta, float ln_eps, float *out);

/* The per-step activations of the Gaussian parameters as render() consumes them (/root/reference/scene_reconstruction/
 * gaussian_model.py:96-121 via gaussian_renderer/__init__.py:92-118): opacity[P] = sigmoid(opacity_raw), scales[P][3] = exp(scaling_raw),
 * shs[P][16][3] = cat(features_dc[P][1][3], features_rest[P][15][3]) in one launch, and their backward in one launch (any incoming
 * gradient may be NULL = zero). */
int csplat_gauss_act_fwd(void *stream, int64_t P, const float *opacity_raw, const float *scaling_raw, const float *features_dc,
                         const float *features_rest, float *opacity, float *scales, float *shs);
int csplat_gauss_act_bwd(void *stream, int64_t P, const float *opacity, const float *scales, const float *g_opacity,
                         const float *g_scales, const float *g_shs, float *d_opacity_raw, float *d_scaling_raw,
                         float *d_features_dc, float *d_features_rest);

#ifdef __cplusplus
}
#endif
#endif /* CSPLAT_H */
