// csplat_gnn.hip -- MeshNet message-passing data movement for gfx950.
//
// Replaces the torch_geometric MessagePassing machinery inside InteractionNetwork.propagate
// (/root/reference/meshnet/graph_network.py:173-174 gather x_i/x_j, :197 concat, :136 aggr='add';
// SURVEY.md 2.1 K10-K12).  The [E,3L] concat is never materialised: the first edge-MLP layer is split into
// node-level products (xa = x W_i^T, xb = x W_j^T, done by rocBLAS on the Python side) and this file's
// gather-add kernel; the scatter-add is a segmented sum over a CSR-by-destination ordering with a fixed
// (ascending edge id) order, i.e. deterministic and without float atomics.  All kernels are HBM-bound row
// movers: 16 B per lane, a row of L floats is read by L/4 consecutive lanes (full 128-B lines at L >= 32).
#include "csplat_common.h"

namespace {

__global__ __launch_bounds__(256) void k_count(int64_t E, const int64_t *__restrict__ keys, uint32_t *__restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&cnt[keys[e]], 1u);
}
__global__ __launch_bounds__(256) void k_rowptr(int N, const uint32_t *__restrict__ incl, int32_t *__restrict__ rowptr) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) rowptr[0] = 0;
    if (n < N) rowptr[n + 1] = (int32_t)incl[n];
}
__global__ __launch_bounds__(256) void k_fill(int64_t E, const int64_t *__restrict__ keys, const int32_t *__restrict__ rowptr,
                                               uint32_t *__restrict__ cursor, int32_t *__restrict__ perm) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        const int64_t k = keys[e];
        const uint32_t slot = atomicAdd(&cursor[k], 1u);
        perm[rowptr[k] + slot] = (int32_t)e;
    }
}
// each row is put into ascending edge-id order (rows are short: mesh degree), making the summation order fixed
__global__ __launch_bounds__(256) void k_sort_rows(int N, const int32_t *__restrict__ rowptr, int32_t *__restrict__ perm) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int s = rowptr[n], e = rowptr[n + 1];
    for (int i = s + 1; i < e; i++) {
        const int32_t v = perm[i];
        int j = i - 1;
        while (j >= s && perm[j] > v) { perm[j + 1] = perm[j]; j--; }
        perm[j + 1] = v;
    }
}

// K14: the per-step graph features of the rollout (PyG Cartesian(norm=False) + Distance(norm=False) on the current node
// positions, /root/reference/train_meshnet_sim.py:152 `graph = transformer(graph)`): one edge per lane
__global__ __launch_bounds__(256) void k_edge_features(int64_t E, const float *__restrict__ pos, const int64_t *__restrict__ ei,
                                                        float4 *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int64_t r = ei[e], c = ei[E + e];
    const float dx = pos[3 * r] - pos[3 * c], dy = pos[3 * r + 1] - pos[3 * c + 1], dz = pos[3 * r + 2] - pos[3 * c + 2];
    out[e] = make_float4(dx, dy, dz, sqrtf(dx * dx + dy * dy + dz * dz));
}

// (the same features for the edges in ANOTHER order: row r of `out` = edge order[r] -- the rollout encodes its edges in destination order)
// (absmax, or NULL: max |value| of the rows as float bits, by atomicMax -- bits of non-negative floats order as integers; the caller zeroes it)
__global__ __launch_bounds__(256) void k_edge_features_ordered(int64_t E, const float *__restrict__ pos, const int64_t *__restrict__ ei,
                                                                const int64_t *__restrict__ order, float4 *__restrict__ out,
                                                                unsigned *__restrict__ absmax) {
    // (grid-stride over at most 512 workgroups, ONE atomic per workgroup: thousands of atomics on one word serialise at the memory side --
    //  one per wave cost 58 us at E = 300k against 6 us for the features themselves)
    __shared__ float s_m[4];
    float m = 0.f;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < E; q += (int64_t)gridDim.x * 256) {
        const int64_t e = order[q];
        const int64_t r = ei[e], c = ei[E + e];
        const float dx = pos[3 * r] - pos[3 * c], dy = pos[3 * r + 1] - pos[3 * c + 1], dz = pos[3 * r + 2] - pos[3 * c + 2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        out[q] = make_float4(dx, dy, dz, len);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fmaxf(fabsf(dz), len)));
    }
    if (absmax) {
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
            if (m > 0.f) atomicMax(absmax, __float_as_uint(m));
        }
    }
}

// ---- the rollout step's head and tail (round 6: no stock launch is left in a recorded rollout step).  Reference: the feature assembly of
// /root/reference/meshnet/cloth_network.py:72-110 (cat(velocity history, one_hot(node type)) -> node normaliser), the de-normalisation and
// v_next = v[:, -3:] + acc of :163-193, and the loop's pinning / integration / history shift, train_meshnet_sim.py:176,256-262.
// Elementwise arithmetic in the order torch's own ops apply it ((x - mean) / std; y * std + mean; then + v): no contraction.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_rollout_head(int N, int H, int T, const float *__restrict__ hist, const int *__restrict__ node_type,
                                                       const float *__restrict__ mean, const float *__restrict__ stdv,
                                                       float *__restrict__ feats, int *__restrict__ counter, unsigned *__restrict__ absmax) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0 && counter) *counter += 1;            // (the step's number + 1: k_rollout_integrate reads it behind this launch)
    float m = 0.f;
    if (n < N) {
        const int F = 3 * H + T;
        float *o = feats + (size_t)n * F;
        for (int h = 0; h < H; h++)
            for (int c = 0; c < 3; c++) {
                const int j = 3 * h + c;
                const float x = hist[((size_t)h * N + n) * 3 + c];
                const float y = mean ? (x - mean[j]) / stdv[j] : x;
                o[j] = y;
                m = fmaxf(m, fabsf(y));
            }
        const int ty = node_type[n];
        for (int t = 0; t < T; t++) {
            const int j = 3 * H + t;
            const float x = ty == t ? 1.f : 0.f;
            const float y = mean ? (x - mean[j]) / stdv[j] : x;
            o[j] = y;
            m = fmaxf(m, fabsf(y));
        }
    }
    if (absmax) {       // (max |feature| as float bits: what the encoder's fp16 pieces are scaled by; zeroed by k_rollout_integrate / the caller)
        __shared__ float s_m[4];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
            if (m > 0.f) atomicMax(absmax, __float_as_uint(m));
        }
    }
}
// decoder's last Linear (128 -> D <= 4) + de-normalisation + v_next = last velocity + acceleration; *fine is cleared by a non-finite row.
// Half a wave per row: lane q of the half reads columns 4q .. 4q+3 (one coalesced 512-byte row per half-wave; a thread per row read 64
// different lines per instruction: 16 us at N = 10^4), the D partial sums cross the half by five xor-shuffles -- every lane ends with the
// same totals in the same order.
__global__ __launch_bounds__(256) void k_rollout_decode(int N, int D, const float *__restrict__ h, const float *__restrict__ W,
                                                         const float *__restrict__ b, const float *__restrict__ omean,
                                                         const float *__restrict__ ostd, const float *__restrict__ last_v,
                                                         float *__restrict__ v, int *__restrict__ fine) {
    const int n = (blockIdx.x * 256 + threadIdx.x) >> 5, q = threadIdx.x & 31;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4 *>(h + (size_t)n * 128)[q];
    float y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; d++)
        if (d < D) {
            const float4 w = reinterpret_cast<const float4 *>(W + (size_t)d * 128)[q];
            float t = a.x * w.x; t = t + a.y * w.y; t = t + a.z * w.z; t = t + a.w * w.w;
            for (int o = 16; o > 0; o >>= 1) t = t + __shfl_xor(t, o, 32);
            y[d] = t;
        }
    bool ok = true;
    for (int d = 0; d < D; d++) {
        float acc = y[d] + b[d];
        if (omean) acc = acc * ostd[d] + omean[d];
        const float r = last_v[(size_t)n * D + d] + acc;
        if (q == 0) v[(size_t)n * D + d] = r;
        ok = ok && (r - r == 0.f);
    }
    if (!ok && q == 0) *fine = 0;
}
// pin the grasped node to the step's action, leave the step's row of the predictions, integrate, shift the velocity history
__global__ __launch_bounds__(256) void k_rollout_integrate(int N, int H, int D, float *__restrict__ v, const float *__restrict__ actions,
                                                            const int *__restrict__ counter, long long grasped, float *__restrict__ pos,
                                                            float *__restrict__ hist, float *__restrict__ preds, unsigned *__restrict__ absmax2) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0 && absmax2) { absmax2[0] = 0u; absmax2[1] = 0u; }      // (the step's two absmax words, consumed by its encoders: zero for the next step)
    if (n >= N) return;
    const int k = *counter - 1;
    for (int d = 0; d < D; d++) {
        const size_t j = (size_t)n * D + d;
        const float val = (long long)n == grasped ? actions[(size_t)k * D + d] : v[j];
        v[j] = val;
        preds[(size_t)k * N * D + j] = val;
        pos[j] = pos[j] + val;
        for (int h = 0; h + 1 < H; h++) hist[(size_t)h * N * D + j] = hist[(size_t)(h + 1) * N * D + j];
        hist[(size_t)(H - 1) * N * D + j] = val;
    }
}
#pragma clang fp contract(fast)

// row movers are templated on the per-lane vector: float4 (16 B/lane) when L % 4 == 0, float otherwise
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vrelu(float4 a) { return make_float4(relu_keep_nan(a.x), relu_keep_nan(a.y), relu_keep_nan(a.z), relu_keep_nan(a.w)); }
__device__ __forceinline__ float vrelu(float a) { return relu_keep_nan(a); }
__device__ __forceinline__ float4 vmask(float4 g, float4 o) {
    return make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
}
__device__ __forceinline__ float vmask(float g, float o) { return o > 0.f ? g : 0.f; }
__device__ __forceinline__ void vzero(float4 &a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vzero(float &a) { a = 0.f; }

template <typename VT>
__global__ __launch_bounds__(256) void k_edge_combine_fwd(int64_t E, int LV, const int64_t *__restrict__ ei,
                                                           const VT *__restrict__ xa, const VT *__restrict__ xb,
                                                           const VT *__restrict__ ec, int relu, VT *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * LV) return;
    const int64_t e = t / LV;
    const int c = (int)(t - e * LV);
    const int64_t src = ei[e], dst = ei[E + e];
    VT r = vadd(vadd(xa[dst * LV + c], xb[src * LV + c]), ec[t]);
    if (relu) r = vrelu(r);
    out[t] = r;
}

template <typename VT>
__global__ __launch_bounds__(256) void k_relu_mask(int64_t nv, const VT *__restrict__ g, const VT *__restrict__ out,
                                                    VT *__restrict__ gm) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nv) gm[t] = vmask(g[t], out[t]);
}

// agg[n][c] = sum over the row's edges, ascending edge id, plain sequential fp32 adds (== index_add_ on the CPU)
// compensated (Neumaier) accumulation: acc carries the running fp32 sum, comp the rounding errors of every add; acc + comp is
// the exact sum to within ~1 ulp whatever the number of terms -- a hub node of an irregular graph collects thousands of
// messages, where a plain fp32 running sum is off by n * eps (1.5e-4 at n = 2500) and that error then rides through the
// whole backward.  Order: ascending edge id (fixed), so the result is deterministic.
__device__ __forceinline__ void kadd(float &acc, float &comp, float v) {
#pragma clang fp contract(off)
    const float t = acc + v;
    comp += fabsf(acc) >= fabsf(v) ? (acc - t) + v : (v - t) + acc;
    acc = t;
}
__device__ __forceinline__ void kadd(float4 &acc, float4 &comp, float4 v) {
    kadd(acc.x, comp.x, v.x); kadd(acc.y, comp.y, v.y); kadd(acc.z, comp.z, v.z); kadd(acc.w, comp.w, v.w);
}
template <typename VT>
__global__ __launch_bounds__(256) void k_segment_sum(int N, int LV, const VT *__restrict__ msg,
                                                      const int32_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                                      VT *__restrict__ agg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * LV) return;
    const int n = (int)(t / LV), c = (int)(t - (int64_t)n * LV);
    const int s = rowptr[n], e = rowptr[n + 1];
    VT acc, comp;
    vzero(acc); vzero(comp);
    int i = s;
    for (; i + 4 <= e; i += 4) {  // 4 independent row loads in flight per lane, summed in list order
        const int p0 = perm[i], p1 = perm[i + 1], p2 = perm[i + 2], p3 = perm[i + 3];
        const VT v0 = msg[(int64_t)p0 * LV + c], v1 = msg[(int64_t)p1 * LV + c], v2 = msg[(int64_t)p2 * LV + c],
                 v3 = msg[(int64_t)p3 * LV + c];
        kadd(acc, comp, v0); kadd(acc, comp, v1); kadd(acc, comp, v2); kadd(acc, comp, v3);
    }
    for (; i < e; i++) kadd(acc, comp, msg[(int64_t)perm[i] * LV + c]);
    agg[t] = vadd(acc, comp);
}

template <typename VT>
__global__ __launch_bounds__(256) void k_gather_rows(int64_t E, int LV, const VT *__restrict__ rows,
                                                      const int64_t *__restrict__ keys, VT *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * LV) return;
    const int64_t e = t / LV;
    const int c = (int)(t - e * LV);
    out[t] = rows[keys[e] * LV + c];
}

// the same pass also leaves max |value| (bits of a non-negative float order as integers; *amax zeroed by the caller).  Grid-stride: a
// wave carries its maximum through all its elements and touches the one shared word once, and only if it beats what is there (a stale
// read costs an extra atomic, never the result) -- one atomic per 256 elements on one address took 1.7 ms at E = 3e5
__global__ __launch_bounds__(256) void k_gather_rows_absmax(int64_t E, int LV, const float4 *__restrict__ rows, const int64_t *__restrict__ keys,
                                                             float4 *__restrict__ out, unsigned *__restrict__ amax) {
    float m = 0.f;
    const int64_t total = E * LV;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = t / LV;
        const float4 v = rows[keys[e] * LV + (int)(t - e * LV)];
        out[t] = v;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > *reinterpret_cast<volatile unsigned *>(amax)) atomicMax(amax, __float_as_uint(m));
}

// ---- the rollout's `real_world` refinement (/root/reference/train_meshnet_sim.py:211-250): per rollout step ten Adam iterations on the
// predicted velocities v, loss = sum_e w_e (|(x + v)[row_e] - (x + v)[col_e]| - L0_e)^2 (w_e = 0 for the entry the reference zeroes with
// `length_deviation[grasped_particle] *= 0`).  ONE launch per iteration, node-centric over the graph's two CSR orderings: thread i forms
// dL/dv_i from its out- and in-edges (every edge length is evaluated from either end -- cheaper than six float atomics per edge, and
// bit-reproducible) and takes the Adam step of its own three coordinates in the same pass (torch.optim.Adam's arithmetic order, as
// csplat_optim.hip).  v is double-buffered by the caller: neighbours are read from v_in, v_out is written.
__global__ __launch_bounds__(256) void k_edge_len_adam(int N, long long E, const float *__restrict__ pos, const float *__restrict__ v_in,
                                                       float *__restrict__ v_out, float *__restrict__ m, float *__restrict__ sq,
                                                       const int64_t *__restrict__ ei, const float *__restrict__ rest_len,
                                                       const float *__restrict__ edge_w, const int *__restrict__ dst_rowptr,
                                                       const int *__restrict__ dst_perm, const int *__restrict__ src_rowptr,
                                                       const int *__restrict__ src_perm, float w1, float w2, float b2, float eps,
                                                       float bc2_sqrt, float step_size) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t i3 = 3 * (size_t)i;
    const float vx = v_in[i3], vy = v_in[i3 + 1], vz = v_in[i3 + 2];
    const float px = pos[i3] + vx, py = pos[i3 + 1] + vy, pz = pos[i3 + 2] + vz;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int e = src_rowptr[i], end = src_rowptr[i + 1]; e < end; e++) {     // edges with row == i: d = p_i - p_col, dL/dp_i = +2 dev d / len
        const int id = src_perm[e];
        const size_t b = 3 * (size_t)ei[E + id];
        const float dx = px - (pos[b] + v_in[b]), dy = py - (pos[b + 1] + v_in[b + 1]), dz = pz - (pos[b + 2] + v_in[b + 2]);
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float dev = (len - rest_len[id]) * (edge_w ? edge_w[id] : 1.f);
        const float k = len > 0.f ? 2.f * dev / len : 0.f;                    // (torch.norm's backward: zero at a zero-length edge)
        gx += k * dx; gy += k * dy; gz += k * dz;
    }
    for (int e = dst_rowptr[i], end = dst_rowptr[i + 1]; e < end; e++) {     // edges with col == i: d = p_row - p_i, dL/dp_i = -2 dev d / len
        const int id = dst_perm[e];
        const size_t a = 3 * (size_t)ei[id];
        const float dx = (pos[a] + v_in[a]) - px, dy = (pos[a + 1] + v_in[a + 1]) - py, dz = (pos[a + 2] + v_in[a + 2]) - pz;
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float dev = (len - rest_len[id]) * (edge_w ? edge_w[id] : 1.f);
        const float k = len > 0.f ? 2.f * dev / len : 0.f;
        gx -= k * dx; gy -= k * dy; gz -= k * dz;
    }
    const float g[3] = {gx, gy, gz}, v0[3] = {vx, vy, vz};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float mm = m[i3 + c], ss = sq[i3 + c];
        mm = mm + (g[c] - mm) * w1;                         // exp_avg.lerp_(grad, 1 - beta1)
        ss = ss * b2 + w2 * g[c] * g[c];                    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(ss) / bc2_sqrt + eps;
        v_out[i3 + c] = v0[c] - step_size * (mm / denom);
        m[i3 + c] = mm; sq[i3 + c] = ss;
    }
}
}  // namespace

extern "C" {

int csplat_gnn_edge_features(void *stream, int64_t E, const float *pos, const int64_t *edge_index, float *out) {
    CSPLAT_REQUIRE(E >= 0 && (E == 0 || (pos && edge_index && out)), "csplat_gnn_edge_features: bad arguments");
    CSPLAT_REQUIRE(((uintptr_t)out & 15u) == 0, "csplat_gnn_edge_features: out must be 16-byte aligned");
    if (E == 0) return 0;
    k_edge_features<<<cdiv(E, 256), 256, 0, (hipStream_t)stream>>>(E, pos, edge_index, (float4 *)out);
    LAUNCH_CHECK();
    return 0;
}
int csplat_gnn_edge_features_ordered(void *stream, int64_t E, const float *pos, const int64_t *edge_index, const int64_t *order, float *out,
                                     float *absmax) {
    CSPLAT_REQUIRE(E >= 0 && (E == 0 || (pos && edge_index && order && out)), "csplat_gnn_edge_features_ordered: bad arguments");
    CSPLAT_REQUIRE(((uintptr_t)out & 15u) == 0, "csplat_gnn_edge_features_ordered: out must be 16-byte aligned");
    if (E == 0) return 0;
    const int64_t nb = cdiv(E, 256);
    k_edge_features_ordered<<<(int)(nb < 512 ? nb : 512), 256, 0, (hipStream_t)stream>>>(E, pos, edge_index, order, (float4 *)out, (unsigned *)absmax);
    LAUNCH_CHECK();
    return 0;
}
int csplat_rollout_head(void *stream, int N, int H, int T, const float *hist, const int32_t *node_type, const float *mean, const float *stdv,
                        float *feats, int32_t *counter, float *absmax) {
    CSPLAT_REQUIRE(N >= 0 && H >= 1 && H <= 16 && T >= 0 && T <= 16 && (N == 0 || (hist && node_type && feats)) && ((mean == nullptr) == (stdv == nullptr)),
                   "csplat_rollout_head: bad arguments");
    k_rollout_head<<<cdiv(N > 0 ? N : 1, 256), 256, 0, (hipStream_t)stream>>>(N, H, T, hist, node_type, mean, stdv, feats, counter, (unsigned *)absmax);
    LAUNCH_CHECK();
    return 0;
}
int csplat_rollout_decode(void *stream, int N, int D, const float *h, const float *W, const float *b, const float *omean, const float *ostd,
                          const float *last_v, float *v, int32_t *fine) {
    CSPLAT_REQUIRE(N >= 0 && D >= 1 && D <= 4 && (N == 0 || (h && W && b && last_v && v && fine)) && ((omean == nullptr) == (ostd == nullptr)),
                   "csplat_rollout_decode: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)h | (uintptr_t)W) & 15u) == 0, "csplat_rollout_decode: h and W must be 16-byte aligned");
    if (N == 0) return 0;
    k_rollout_decode<<<cdiv((int64_t)N * 32, 256), 256, 0, (hipStream_t)stream>>>(N, D, h, W, b, omean, ostd, last_v, v, fine);
    LAUNCH_CHECK();
    return 0;
}
int csplat_rollout_integrate(void *stream, int N, int H, int D, float *v, const float *actions, const int32_t *counter, int64_t grasped,
                             float *pos, float *hist, float *preds, float *absmax2) {
    CSPLAT_REQUIRE(N >= 0 && H >= 1 && D >= 1 && D <= 4 && (N == 0 || (v && actions && counter && pos && hist && preds)), "csplat_rollout_integrate: bad arguments");
    if (N == 0) return 0;
    k_rollout_integrate<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(N, H, D, v, actions, counter, (long long)grasped, pos, hist, preds, (unsigned *)absmax2);
    LAUNCH_CHECK();
    return 0;
}

// The `real_world` refinement of one rollout step (/root/reference/train_meshnet_sim.py:211-250): `iters` Adam iterations (a fresh
// torch.optim.Adam(lr): zero moments, step counts 1..iters) on v [N][3] against the squared deviation of the edge lengths of pos + v from
// rest_len [E]; edge_w [E] or NULL weights the deviations (0 for the entry the reference zeroes).  v is updated IN PLACE; scratch =
// 9 N floats (the other half of the double buffer and both moments; contents irrelevant on entry).  edge_index [2][E] int64; the two
// CSR orderings as csplat_gnn_build_csr leaves them.  Nothing is read back: recordable into a hipGraph.
int csplat_gnn_edge_length_refine(void *stream, int N, int64_t E, const float *pos, float *v, const int64_t *edge_index, const float *rest_len,
                                  const float *edge_w, const int32_t *dst_rowptr, const int32_t *dst_perm, const int32_t *src_rowptr,
                                  const int32_t *src_perm, int iters, double lr, double beta1, double beta2, double eps, float *scratch) {
    CSPLAT_REQUIRE(N >= 0 && E >= 0 && iters >= 0 && iters <= 1000, "csplat_gnn_edge_length_refine: bad sizes");
    if (N == 0 || iters == 0) return 0;
    CSPLAT_REQUIRE(pos && v && scratch && dst_rowptr && src_rowptr && (E == 0 || (edge_index && rest_len && dst_perm && src_perm)),
                   "csplat_gnn_edge_length_refine: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    float *v2 = scratch, *m = scratch + 3 * (size_t)N, *sq = scratch + 6 * (size_t)N;
    HIP_TRY(hipMemsetAsync(m, 0, 6 * (size_t)N * sizeof(float), s));
    float *cur = v, *nxt = v2;
    for (int t = 1; t <= iters; t++) {
        const double bc1 = 1.0 - pow(beta1, (double)t);
        const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)t));
        k_edge_len_adam<<<cdiv(N, 256), 256, 0, s>>>(N, (long long)E, pos, cur, nxt, m, sq, edge_index, rest_len, edge_w, dst_rowptr, dst_perm,
                                                     src_rowptr, src_perm, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)beta2, (float)eps,
                                                     bc2_sqrt, (float)(lr / bc1));
        LAUNCH_CHECK();
        float *t_ = cur; cur = nxt; nxt = t_;
    }
    if (cur != v) HIP_TRY(hipMemcpyAsync(v, cur, 3 * (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

size_t csplat_gnn_csr_temp_bytes(int N, int64_t E) {
    (void)E;
    return 2 * align256((size_t)(N + 1) * 4) + csplat_scan_temp_bytes(N);
}

int csplat_gnn_build_csr(void *stream, int N, int64_t E, const int64_t *keys, int32_t *rowptr, int32_t *perm, void *temp) {
    hipStream_t s = (hipStream_t)stream;
    CSPLAT_REQUIRE(N > 0 && E >= 0 && E < (int64_t)1 << 31, "csplat_gnn_build_csr: bad sizes");
    uint32_t *cnt = (uint32_t *)temp;
    uint32_t *incl = (uint32_t *)((char *)temp + align256((size_t)(N + 1) * 4));
    void *scan_tmp = (char *)temp + 2 * align256((size_t)(N + 1) * 4);
    HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)N * 4, s));
    if (E > 0) { k_count<<<cdiv(E, 256), 256, 0, s>>>(E, keys, cnt); LAUNCH_CHECK(); }
    if (int rc = csplat_inclusive_scan_u32(s, cnt, incl, N, scan_tmp)) return rc;
    k_rowptr<<<cdiv(N, 256), 256, 0, s>>>(N, incl, rowptr);
    LAUNCH_CHECK();
    HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)N * 4, s));
    if (E > 0) {
        k_fill<<<cdiv(E, 256), 256, 0, s>>>(E, keys, rowptr, cnt, perm);
        LAUNCH_CHECK();
        k_sort_rows<<<cdiv(N, 256), 256, 0, s>>>(N, rowptr, perm);
        LAUNCH_CHECK();
    }
    return 0;
}

int csplat_gnn_edge_combine_fwd(void *stream, int N, int64_t E, int L, const int64_t *edge_index, const float *xa,
                                const float *xb, const float *ec, int relu, float *out) {
    (void)N;
    CSPLAT_REQUIRE(L > 0, "latent width must be positive");
    if (E == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    if (L % 4 == 0)
        k_edge_combine_fwd<float4><<<cdiv(E * (L / 4), 256), 256, 0, s>>>(E, L / 4, edge_index, (const float4 *)xa,
                                                                          (const float4 *)xb, (const float4 *)ec, relu,
                                                                          (float4 *)out);
    else
        k_edge_combine_fwd<float><<<cdiv(E * L, 256), 256, 0, s>>>(E, L, edge_index, xa, xb, ec, relu, out);
    LAUNCH_CHECK();
    return 0;
}

int csplat_gnn_segment_sum(void *stream, int N, int64_t E, int L, const float *msg, const int32_t *rowptr,
                           const int32_t *perm, float *agg) {
    (void)E;
    CSPLAT_REQUIRE(L > 0, "latent width must be positive");
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    if (L % 4 == 0)
        k_segment_sum<float4><<<cdiv((int64_t)N * (L / 4), 256), 256, 0, s>>>(N, L / 4, (const float4 *)msg, rowptr, perm,
                                                                              (float4 *)agg);
    else
        k_segment_sum<float><<<cdiv((int64_t)N * L, 256), 256, 0, s>>>(N, L, msg, rowptr, perm, agg);
    LAUNCH_CHECK();
    return 0;
}

int csplat_gnn_edge_combine_bwd(void *stream, int N, int64_t E, int L, const float *g, const float *out, int relu,
                                const int32_t *rowptr_dst, const int32_t *perm_dst, const int32_t *rowptr_src,
                                const int32_t *perm_src, float *g_masked, float *dxa, float *dxb) {
    CSPLAT_REQUIRE(L > 0, "latent width must be positive");
    hipStream_t s = (hipStream_t)stream;
    const float *gm = g;
    if (relu) {
        if (E > 0) {
            ProfScope ps(PROF_GNN, s);
            if (L % 4 == 0)
                k_relu_mask<float4><<<cdiv(E * (L / 4), 256), 256, 0, s>>>(E * (L / 4), (const float4 *)g,
                                                                           (const float4 *)out, (float4 *)g_masked);
            else
                k_relu_mask<float><<<cdiv(E * L, 256), 256, 0, s>>>(E * L, g, out, g_masked);
            LAUNCH_CHECK();
        }
        gm = g_masked;
    }
    if (int rc = csplat_gnn_segment_sum(stream, N, E, L, gm, rowptr_dst, perm_dst, dxa)) return rc;
    return csplat_gnn_segment_sum(stream, N, E, L, gm, rowptr_src, perm_src, dxb);
}

int csplat_gnn_gather_rows(void *stream, int64_t E, int L, const float *rows, const int64_t *keys, float *out) {
    CSPLAT_REQUIRE(L > 0, "latent width must be positive");
    if (E == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    if (L % 4 == 0)
        k_gather_rows<float4><<<cdiv(E * (L / 4), 256), 256, 0, s>>>(E, L / 4, (const float4 *)rows, keys, (float4 *)out);
    else
        k_gather_rows<float><<<cdiv(E * L, 256), 256, 0, s>>>(E, L, rows, keys, out);
    LAUNCH_CHECK();
    return 0;
}

int csplat_gnn_gather_rows_absmax(void *stream, int64_t E, int L, const float *rows, const int64_t *keys, float *out, float *absmax) {
    CSPLAT_REQUIRE(E >= 0 && L > 0 && L % 4 == 0 && absmax && (E == 0 || (rows && keys && out)), "csplat_gnn_gather_rows_absmax: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)rows | (uintptr_t)out) & 15u) == 0, "csplat_gnn_gather_rows_absmax: rows / out must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(absmax, 0, sizeof(float), s));
    if (E == 0) return 0;
    ProfScope ps(PROF_GNN, s);
    const int64_t nb = cdiv(E * (L / 4), 256);
    k_gather_rows_absmax<<<(int)(nb < 4096 ? nb : 4096), 256, 0, s>>>(E, L / 4, (const float4 *)rows, keys, (float4 *)out, (unsigned *)absmax);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// ---- LayerNorm over 128-wide rows for the TRAINING path of the MeshNet MLPs (nn.LayerNorm(128) after every edge / node MLP,
// /root/reference/meshnet/graph_network.py:86-97,139-150).  On [E = 3e5, 128] rows the library kernels take 165 us forward and
// 170 + 177 us backward (two passes: gamma/beta partials, then the input gradient); these are single HBM passes:
//   forward  y = (x - mean) * rstd * gamma + beta, stats[row] = (mean, rstd)               read 512 B + write 520 B per row
//   backward dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat)), and per-workgroup partial column sums of
//            g*xhat (dgamma) and g (dbeta), reduced in a fixed order by a second tiny launch (deterministic)
// A row = 32 lanes x float4; a wave holds two rows; row statistics by 5 xor-shuffles inside the 32-lane half.
namespace {
constexpr int LN_THREADS = 256, LN_ROWS_PER_BLOCK_ITER = LN_THREADS / 32;
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(LN_THREADS) void k_ln128_fwd(int64_t M, const float4 *__restrict__ x, const float4 *__restrict__ gamma,
                                                           const float4 *__restrict__ beta, float eps, float4 *__restrict__ y,
                                                           float2 *__restrict__ stats) {
    const int sub = threadIdx.x & 31;
    const float4 ga = gamma[sub], be = beta[sub];
    for (int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK_ITER + (threadIdx.x >> 5); row < M; row += (int64_t)gridDim.x * LN_ROWS_PER_BLOCK_ITER) {
        const float4 v = x[row * 32 + sub];
        const float mean = half_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 128.f);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        const float var = half_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.f / 128.f);
        const float rstd = rsqrtf(var + eps);
        y[row * 32 + sub] = make_float4(d0 * rstd * ga.x + be.x, d1 * rstd * ga.y + be.y, d2 * rstd * ga.z + be.z, d3 * rstd * ga.w + be.w);
        if (sub == 0) stats[row] = make_float2(mean, rstd);
    }
}
__global__ __launch_bounds__(LN_THREADS) void k_ln128_bwd(int64_t M, const float4 *__restrict__ g, const float4 *__restrict__ x,
                                                           const float2 *__restrict__ stats, const float4 *__restrict__ gamma,
                                                           float4 *__restrict__ dx, float4 *__restrict__ part_gamma,
                                                           float4 *__restrict__ part_beta, float4 *__restrict__ part_dx,
                                                           const int64_t *__restrict__ g_rows, int x_normalized) {
    __shared__ float4 s_g[LN_THREADS], s_b[LN_THREADS], s_x[LN_THREADS];
    const int sub = threadIdx.x & 31;
    const float4 ga = gamma[sub];
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag, ax = ag;
    for (int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK_ITER + (threadIdx.x >> 5); row < M; row += (int64_t)gridDim.x * LN_ROWS_PER_BLOCK_ITER) {
        const float4 gv = g[(g_rows ? g_rows[row] : row) * 32 + sub], xv = x[row * 32 + sub];
        const float2 st = stats[row];
        const float sub_mean = x_normalized ? 0.f : st.x, mul = x_normalized ? 1.f : st.y;   // x_normalized: x already is xhat
        const float h0 = (xv.x - sub_mean) * mul, h1 = (xv.y - sub_mean) * mul, h2 = (xv.z - sub_mean) * mul, h3 = (xv.w - sub_mean) * mul;
        const float w0 = gv.x * ga.x, w1 = gv.y * ga.y, w2 = gv.z * ga.z, w3 = gv.w * ga.w;
        const float m1 = half_sum((w0 + w1) + (w2 + w3)) * (1.f / 128.f);
        const float m2 = half_sum((w0 * h0 + w1 * h1) + (w2 * h2 + w3 * h3)) * (1.f / 128.f);
        const float4 dv = make_float4(st.y * (w0 - m1 - h0 * m2), st.y * (w1 - m1 - h1 * m2), st.y * (w2 - m1 - h2 * m2), st.y * (w3 - m1 - h3 * m2));
        dx[row * 32 + sub] = dv;
        ax.x += dv.x; ax.y += dv.y; ax.z += dv.z; ax.w += dv.w;
        ag.x += gv.x * h0; ag.y += gv.y * h1; ag.z += gv.z * h2; ag.w += gv.w * h3;
        ab.x += gv.x; ab.y += gv.y; ab.z += gv.z; ab.w += gv.w;
    }
    s_g[threadIdx.x] = ag; s_b[threadIdx.x] = ab; s_x[threadIdx.x] = ax;
    __syncthreads();
    if (threadIdx.x < 32) {   // the block's 8 row-slots, in order
        float4 tg = s_g[threadIdx.x], tb = s_b[threadIdx.x], tx = s_x[threadIdx.x];
        for (int k = 1; k < LN_ROWS_PER_BLOCK_ITER; k++) {
            const float4 a = s_g[k * 32 + threadIdx.x], b = s_b[k * 32 + threadIdx.x], c = s_x[k * 32 + threadIdx.x];
            tg.x += a.x; tg.y += a.y; tg.z += a.z; tg.w += a.w; tb.x += b.x; tb.y += b.y; tb.z += b.z; tb.w += b.w;
            tx.x += c.x; tx.y += c.y; tx.z += c.z; tx.w += c.w;
        }
        part_gamma[(size_t)blockIdx.x * 32 + threadIdx.x] = tg;
        part_beta[(size_t)blockIdx.x * 32 + threadIdx.x] = tb;
        if (part_dx) part_dx[(size_t)blockIdx.x * 32 + threadIdx.x] = tx;
    }
}
// column sums of [nblocks][128] partials: 8 slices of the block range summed in parallel (ascending inside a slice), then the
// 8 slice sums in order -- a fixed association, independent of scheduling
__global__ __launch_bounds__(1024) void k_colsum128(int nblocks, const float *__restrict__ pa, const float *__restrict__ pb,
                                                     float *__restrict__ oa, float *__restrict__ ob) {
    __shared__ float s_part[8][128];
    const float *p = blockIdx.x == 0 ? pa : pb;
    float *o = blockIdx.x == 0 ? oa : ob;
    if (!p || !o) return;
    const int col = threadIdx.x & 127, slice = threadIdx.x >> 7;
    const int per = (nblocks + 7) / 8, b0 = slice * per, b1 = min(nblocks, b0 + per);
    float acc = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {   // 4 independent loads in flight
        const float v0 = p[(size_t)b * 128 + col], v1 = p[(size_t)(b + 1) * 128 + col], v2 = p[(size_t)(b + 2) * 128 + col],
                    v3 = p[(size_t)(b + 3) * 128 + col];
        acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; b < b1; b++) acc += p[(size_t)b * 128 + col];
    s_part[slice][col] = acc;
    __syncthreads();
    if (slice == 0) {
        float t = s_part[0][col];
        for (int k = 1; k < 8; k++) t += s_part[k][col];
        o[col] = t;
    }
}
// g_out = relu_mask(g, out); partial column sums of g_out (the bias gradient) per workgroup: the ReLU backward and the bias
// gradient of a Linear + ReLU layer in one pass over the [E, 128] gradient
__global__ __launch_bounds__(LN_THREADS) void k_relu_mask_bias128(int64_t M, const float4 *__restrict__ g, const float4 *__restrict__ out,
                                                                   float4 *__restrict__ gm, float4 *__restrict__ part_bias) {
    __shared__ float4 s_b[LN_THREADS];
    const int sub = threadIdx.x & 31;
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK_ITER + (threadIdx.x >> 5); row < M; row += (int64_t)gridDim.x * LN_ROWS_PER_BLOCK_ITER) {
        float4 gv = g[row * 32 + sub];
        if (out) { const float4 o = out[row * 32 + sub]; gv = vmask(gv, o); }
        if (gm) gm[row * 32 + sub] = gv;
        ab.x += gv.x; ab.y += gv.y; ab.z += gv.z; ab.w += gv.w;
    }
    s_b[threadIdx.x] = ab;
    __syncthreads();
    if (threadIdx.x < 32) {
        float4 tb = s_b[threadIdx.x];
        for (int k = 1; k < LN_ROWS_PER_BLOCK_ITER; k++) { const float4 b = s_b[k * 32 + threadIdx.x]; tb.x += b.x; tb.y += b.y; tb.z += b.z; tb.w += b.w; }
        part_bias[(size_t)blockIdx.x * 32 + threadIdx.x] = tb;
    }
}
int ln_blocks(int64_t M) { const int64_t want = (M + 63) / 64; return (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want)); }
}  // namespace

extern "C" {
size_t csplat_ln128_partial_floats(int64_t M) { return (size_t)ln_blocks(M) * 128; }

int csplat_ln128_fwd(void *stream, int64_t M, const float *x, const float *gamma, const float *beta, float eps, float *y, float *stats) {
    CSPLAT_REQUIRE(M >= 0 && (M == 0 || (x && gamma && beta && y && stats)), "csplat_ln128_fwd: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15u) == 0 && ((uintptr_t)stats & 7u) == 0,
                   "csplat_ln128_fwd: 16-byte aligned rows");
    if (M == 0) return 0;
    k_ln128_fwd<<<ln_blocks(M), LN_THREADS, 0, (hipStream_t)stream>>>(M, (const float4 *)x, (const float4 *)gamma, (const float4 *)beta, eps,
                                                                     (float4 *)y, (float2 *)stats);
    LAUNCH_CHECK();
    return 0;
}

int csplat_ln128_bwd(void *stream, int64_t M, const float *g, const float *x, const float *stats, const float *gamma, float *dx,
                     float *dgamma, float *dbeta, float *dxsum, const int64_t *g_rows, int x_normalized, float *partials) {
    CSPLAT_REQUIRE(M >= 0 && (M == 0 || (g && x && stats && gamma && dx && dgamma && dbeta && partials)), "csplat_ln128_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        HIP_TRY(hipMemsetAsync(dgamma, 0, 512, s)); HIP_TRY(hipMemsetAsync(dbeta, 0, 512, s));
        if (dxsum) HIP_TRY(hipMemsetAsync(dxsum, 0, 512, s));
        return 0;
    }
    const int nb = ln_blocks(M);
    float *pg = partials, *pb = partials + (size_t)nb * 128, *px = dxsum ? partials + (size_t)2 * nb * 128 : nullptr;
    k_ln128_bwd<<<nb, LN_THREADS, 0, s>>>(M, (const float4 *)g, (const float4 *)x, (const float2 *)stats, (const float4 *)gamma, (float4 *)dx,
                                          (float4 *)pg, (float4 *)pb, (float4 *)px, g_rows, x_normalized);
    LAUNCH_CHECK();
    k_colsum128<<<2, 1024, 0, s>>>(nb, pg, pb, dgamma, dbeta);
    LAUNCH_CHECK();
    if (dxsum) {
        k_colsum128<<<1, 1024, 0, s>>>(nb, px, nullptr, dxsum, nullptr);
        LAUNCH_CHECK();
    }
    return 0;
}

/* gm = out > 0 ? g : 0 (out NULL: gm = g; gm NULL: not written), dbias[c] = sum_rows gm[row][c]; partials: csplat_ln128_partial_floats(M) floats */
int csplat_relu_mask_bias128(void *stream, int64_t M, const float *g, const float *out, float *gm, float *dbias, float *partials) {
    CSPLAT_REQUIRE(M >= 0 && (M == 0 || (g && dbias && partials)), "csplat_relu_mask_bias128: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) { HIP_TRY(hipMemsetAsync(dbias, 0, 512, s)); return 0; }
    const int nb = ln_blocks(M);
    k_relu_mask_bias128<<<nb, LN_THREADS, 0, s>>>(M, (const float4 *)g, (const float4 *)out, (float4 *)gm, (float4 *)partials);
    LAUNCH_CHECK();
    k_colsum128<<<1, 1024, 0, s>>>(nb, partials, nullptr, dbias, nullptr);
    LAUNCH_CHECK();
    return 0;
}
}  // extern "C"
