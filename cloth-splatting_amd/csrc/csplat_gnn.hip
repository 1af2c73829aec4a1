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

// row movers are templated on the per-lane vector: float4 (16 B/lane) when L % 4 == 0, float otherwise
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vrelu(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float vrelu(float a) { return fmaxf(a, 0.f); }
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
template <typename VT>
__global__ __launch_bounds__(256) void k_segment_sum(int N, int LV, const VT *__restrict__ msg,
                                                      const int32_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                                      VT *__restrict__ agg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * LV) return;
    const int n = (int)(t / LV), c = (int)(t - (int64_t)n * LV);
    const int s = rowptr[n], e = rowptr[n + 1];
    VT acc;
    vzero(acc);
    int i = s;
    for (; i + 4 <= e; i += 4) {  // 4 independent row loads in flight per lane, summed in list order
        const int p0 = perm[i], p1 = perm[i + 1], p2 = perm[i + 2], p3 = perm[i + 3];
        const VT v0 = msg[(int64_t)p0 * LV + c], v1 = msg[(int64_t)p1 * LV + c], v2 = msg[(int64_t)p2 * LV + c],
                 v3 = msg[(int64_t)p3 * LV + c];
        acc = vadd(vadd(vadd(vadd(acc, v0), v1), v2), v3);
    }
    for (; i < e; i++) acc = vadd(acc, msg[(int64_t)perm[i] * LV + c]);
    agg[t] = acc;
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

}  // extern "C"
