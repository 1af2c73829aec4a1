// csplat_sim.hip -- output layer of the time-conditioned mesh simulator.
//
// ResidualMeshSimulator (reference meshnet/meshnet_network.py:327-373) ends in Linear(256, 3V): ONE time value feeds the
// whole mesh, so the layer is a matrix-vector product over a [3V, 256] weight (30.7 MB at V = 10k).  A training step
// renders T cameras (t-1, t, t+1), i.e. T such products and, in backward, T outer products into the same weight gradient.
// As M = 1 GEMMs they run far below HBM rate and autograd adds T-1 full-size accumulation passes.  Here the T time rows are
// handled together, the weight is streamed once per direction:
//   forward   y[t][r]  = b[r] + sum_k W[r][k] h[t][k]                       reads W once          (R*K*4 bytes)
//   backward  dW[r][k] = sum_t dy[t][r] h[t][k];  db[r] = sum_t dy[t][r];
//             dh[t][k] = sum_r dy[t][r] W[r][k]                             reads W, writes dW    (2*R*K*4 bytes)
// Both are HBM-bound streams; a wavefront owns a row (64 lanes x float4 = the 256 inputs).  dh is reduced deterministically:
// per-workgroup partials, then a fixed-order sum.
#include "csplat_common.h"

namespace {
constexpr int SIM_K = 256;        // inputs of the layer (hidden width of the simulator MLP)
constexpr int SIM_TMAX = 8;       // time rows per call
constexpr int SIM_BLOCKS = 512;   // 2 workgroups per CU, 4 waves each, 4 rows in flight per wave: 32 KB of loads per CU

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <int T>
__global__ __launch_bounds__(256) void k_rows_dot_fwd(int R, const float4 *__restrict__ W, const float *__restrict__ b,
                                                      const float4 *__restrict__ h, float *__restrict__ y,
                                                      const float *__restrict__ add) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    float4 hr[T];
#pragma unroll
    for (int t = 0; t < T; t++) hr[t] = h[t * 64 + lane];
    for (int r0 = wave * 4; r0 < R; r0 += nwaves * 4) {
        float4 w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = r0 + j < R ? W[(size_t)(r0 + j) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                const float s = wave_sum(dot4(w[j], hr[t]));
                if (lane == 0 && r0 + j < R) {      // (+ add: the simulator's table row, meshnet_network.py:371 `mesh_predictions[t] + residual`)
                    const float v = s + b[r0 + j];
                    y[(size_t)t * R + r0 + j] = add ? add[(size_t)t * R + r0 + j] + v : v;
                }
            }
        }
    }
}

template <int T>
__global__ __launch_bounds__(256) void k_rows_dot_bwd(int R, const float4 *__restrict__ W, const float4 *__restrict__ h,
                                                      const float *__restrict__ dy, float4 *__restrict__ dW,
                                                      float *__restrict__ db, float4 *__restrict__ part) {
    __shared__ float4 s_acc[4][T][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    float4 hr[T], acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        hr[t] = h[t * 64 + lane];
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int r0 = wave * 4; r0 < R; r0 += nwaves * 4) {
        float4 w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = r0 + j < R ? W[(size_t)(r0 + j) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (r0 + j < R) {   // (wave-uniform)
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                float sb = 0.f;
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const float d = dy[(size_t)t * R + r0 + j];
                    sb += d;
                    g.x += d * hr[t].x; g.y += d * hr[t].y; g.z += d * hr[t].z; g.w += d * hr[t].w;
                    acc[t].x += d * w[j].x; acc[t].y += d * w[j].y; acc[t].z += d * w[j].z; acc[t].w += d * w[j].w;
                }
                dW[(size_t)(r0 + j) * 64 + lane] = g;
                if (lane == 0) db[r0 + j] = sb;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; t++) s_acc[wv][t][lane] = acc[t];
    __syncthreads();
    for (int i = threadIdx.x; i < T * 64; i += 256) {
        const int t = i >> 6, l = i & 63;
        float4 a = s_acc[0][t][l];
        for (int k = 1; k < 4; k++) {
            const float4 c = s_acc[k][t][l];
            a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        }
        part[(size_t)blockIdx.x * T * 64 + i] = a;
    }
}

// dh[i] = sum over workgroups (fixed order) of part[g][i], i < T*256.  64 outputs per workgroup, 16 slices of the workgroup
// list summed side by side (a single thread walking all 512 partials is a 40 us chain of dependent-latency loads).
__global__ __launch_bounds__(1024) void k_rows_dot_dh(int n, int groups, const float *__restrict__ part, float *__restrict__ dh) {
    __shared__ float s_part[16][64];
    const int l = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + l;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int g = slice;
        for (; g + 16 < groups; g += 32) {
            s0 += part[(size_t)g * n + i];
            s1 += part[(size_t)(g + 16) * n + i];
        }
        if (g < groups) s0 += part[(size_t)g * n + i];
    }
    s_part[slice][l] = s0 + s1;
    __syncthreads();
    if (slice == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) s += s_part[k][l];
        dh[i] = s;
    }
}

// ---- cloth regularisers of the reconstruction loss (reference scene_reconstruction/train_utils.py:83-102) in one pass.
// D [T][V][3] = deformed vertices of the step's T cameras (t-1, t, t+1):
//   deform-magnitude  lambda_d * 0.5 * (mean_v |D1 - D0|_2 + mean_v |D2 - D1|_2)                    (T >= 3)
//   rigidity          lambda_r * mean_{t,e} | rest_len[e] - |D[t][dst e] - D[t][src e]|_2 |
//   momentum          lambda_m * mean_v |D2 - 2 D1 + D0|_1                                            (T >= 3)
// As torch ops this is ~35 launches forward and ~55 backward of a few microseconds each.  Here: work items 0..V-1 are the
// vertices (node terms), V..V+T*E-1 the (time, edge) pairs; every item adds its loss share to a per-workgroup partial and
// its gradient to grad[T][V][3] (atomics: a vertex collects from ~6 edges + its node terms); the last workgroup to finish
// sums the partials in fixed order -> the loss value is deterministic, the gradient is up to atomic summation order (as
// index_add's is upstream).
__global__ __launch_bounds__(256) void k_cloth_regs(int T, int V, long long E, const float *__restrict__ D,
                                                    const int64_t *__restrict__ ei, const float *__restrict__ rest_len,
                                                    float w_deform, float w_rigid, float w_mom, float *__restrict__ grad,
                                                    float *__restrict__ partial, unsigned int *__restrict__ ticket,
                                                    float *__restrict__ loss) {
    __shared__ float s_red[4];
    __shared__ bool s_last;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n_edge_items = w_rigid != 0.f ? (long long)T * E : 0;
    float acc = 0.f;
    if (i < V) {
        if (T >= 3 && (w_deform != 0.f || w_mom != 0.f)) {
            float d0[3], d1[3], d2[3];
            for (int c = 0; c < 3; c++) {
                d0[c] = D[(size_t)i * 3 + c];
                d1[c] = D[((size_t)V + i) * 3 + c];
                d2[c] = D[((size_t)2 * V + i) * 3 + c];
            }
            float g0[3] = {0.f, 0.f, 0.f}, g1[3] = {0.f, 0.f, 0.f}, g2[3] = {0.f, 0.f, 0.f};
            if (w_deform != 0.f) {   // w_deform = 0.5 * lambda_d / V
                const float a[3] = {d1[0] - d0[0], d1[1] - d0[1], d1[2] - d0[2]};
                const float b[3] = {d2[0] - d1[0], d2[1] - d1[1], d2[2] - d1[2]};
                const float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
                const float nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
                acc += w_deform * (na + nb);
                const float ia = na > 0.f ? w_deform / na : 0.f, ib = nb > 0.f ? w_deform / nb : 0.f;   // d|x|/dx = 0 at x = 0
                for (int c = 0; c < 3; c++) {
                    g0[c] -= ia * a[c];
                    g1[c] += ia * a[c] - ib * b[c];
                    g2[c] += ib * b[c];
                }
            }
            if (w_mom != 0.f) {      // w_mom = lambda_m / V
                for (int c = 0; c < 3; c++) {
                    const float m = d2[c] - 2.f * d1[c] + d0[c];
                    acc += w_mom * fabsf(m);
                    const float sg = m > 0.f ? w_mom : (m < 0.f ? -w_mom : 0.f);
                    g0[c] += sg;
                    g1[c] -= 2.f * sg;
                    g2[c] += sg;
                }
            }
            for (int c = 0; c < 3; c++) {
                atomicAdd(grad + (size_t)i * 3 + c, g0[c]);
                atomicAdd(grad + ((size_t)V + i) * 3 + c, g1[c]);
                atomicAdd(grad + ((size_t)2 * V + i) * 3 + c, g2[c]);
            }
        }
    } else if (i - V < n_edge_items) {   // w_rigid = lambda_r / (T * E)
        const long long k = i - V;
        const int t = (int)(k / E);
        const long long e = k - (long long)t * E;
        const int64_t a = ei[e], b = ei[E + e];   // disp = D[t][ei[1]] - D[t][ei[0]]
        const float *Dt = D + (size_t)t * V * 3;
        const float dx = Dt[b * 3] - Dt[a * 3], dy = Dt[b * 3 + 1] - Dt[a * 3 + 1], dz = Dt[b * 3 + 2] - Dt[a * 3 + 2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float diff = rest_len[e] - len;
        acc += w_rigid * fabsf(diff);
        // d|rest - len| / d len = -sign(rest - len);  d len / d disp = disp / len (0 at len = 0)
        const float sl = diff > 0.f ? -w_rigid : (diff < 0.f ? w_rigid : 0.f);
        const float k_ = len > 0.f ? sl / len : 0.f;
        float *gt = grad + (size_t)t * V * 3;
        atomicAdd(gt + b * 3, k_ * dx); atomicAdd(gt + b * 3 + 1, k_ * dy); atomicAdd(gt + b * 3 + 2, k_ * dz);
        atomicAdd(gt + a * 3, -k_ * dx); atomicAdd(gt + a * 3 + 1, -k_ * dy); atomicAdd(gt + a * 3 + 2, -k_ * dz);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    float s = 0.f;
    for (unsigned j = threadIdx.x; j < gridDim.x; j += 256) s += __builtin_nontemporal_load(partial + j);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        *loss = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        *ticket = 0u;   // ready for the next call
    }
}

// The same terms, node-centric: thread (t, v) visits the edges INTO v (loss + gradient) and OUT OF v (gradient) through the
// static CSR of the cloth graph and writes grad[t][v] once -- no atomics, no zero-fill, every bit reproducible.  Each edge's
// length is evaluated twice (once from either end); that is cheaper than 6 float atomics per (time, edge) on ~V addresses.
__global__ __launch_bounds__(256) void k_cloth_regs_csr(int T, int V, long long E, const float *__restrict__ D,
                                                        const int64_t *__restrict__ ei, const float *__restrict__ rest_len,
                                                        const int *__restrict__ dst_rowptr, const int *__restrict__ dst_perm,
                                                        const int *__restrict__ src_rowptr, const int *__restrict__ src_perm,
                                                        float w_deform, float w_rigid, float w_mom, float *__restrict__ grad,
                                                        float *__restrict__ partial, unsigned int *__restrict__ ticket,
                                                        float *__restrict__ loss) {
    __shared__ float s_red[4];
    __shared__ bool s_last;
    const int v = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    float acc = 0.f;
    if (v < V) {
        const float *Dt = D + (size_t)t * V * 3;
        const float px = Dt[3 * (size_t)v], py = Dt[3 * (size_t)v + 1], pz = Dt[3 * (size_t)v + 2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (w_rigid != 0.f) {
            for (int e = dst_rowptr[v], end = dst_rowptr[v + 1]; e < end; e++) {   // edges (a -> v): disp = D[v] - D[a]
                const int id = dst_perm[e];
                const int64_t a = ei[id];
                const float dx = px - Dt[3 * a], dy = py - Dt[3 * a + 1], dz = pz - Dt[3 * a + 2];
                const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                const float diff = rest_len[id] - len;
                acc += w_rigid * fabsf(diff);
                const float sl = diff > 0.f ? -w_rigid : (diff < 0.f ? w_rigid : 0.f);
                const float k = len > 0.f ? sl / len : 0.f;
                gx += k * dx; gy += k * dy; gz += k * dz;
            }
            for (int e = src_rowptr[v], end = src_rowptr[v + 1]; e < end; e++) {   // edges (v -> b): disp = D[b] - D[v]
                const int id = src_perm[e];
                const int64_t b = ei[E + id];
                const float dx = Dt[3 * b] - px, dy = Dt[3 * b + 1] - py, dz = Dt[3 * b + 2] - pz;
                const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                const float diff = rest_len[id] - len;
                const float sl = diff > 0.f ? -w_rigid : (diff < 0.f ? w_rigid : 0.f);
                const float k = len > 0.f ? sl / len : 0.f;
                gx -= k * dx; gy -= k * dy; gz -= k * dz;
            }
        }
        if (T >= 3 && t < 3 && (w_deform != 0.f || w_mom != 0.f)) {
            float d0[3], d1[3], d2[3], g[3] = {0.f, 0.f, 0.f};
            for (int c = 0; c < 3; c++) {
                d0[c] = D[(size_t)v * 3 + c];
                d1[c] = D[((size_t)V + v) * 3 + c];
                d2[c] = D[((size_t)2 * V + v) * 3 + c];
            }
            if (w_deform != 0.f) {
                const float a[3] = {d1[0] - d0[0], d1[1] - d0[1], d1[2] - d0[2]};
                const float b[3] = {d2[0] - d1[0], d2[1] - d1[1], d2[2] - d1[2]};
                const float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
                const float nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
                if (t == 0) acc += w_deform * (na + nb);
                const float ia = na > 0.f ? w_deform / na : 0.f, ib = nb > 0.f ? w_deform / nb : 0.f;
                for (int c = 0; c < 3; c++) g[c] += t == 0 ? -ia * a[c] : (t == 1 ? ia * a[c] - ib * b[c] : ib * b[c]);
            }
            if (w_mom != 0.f) {
                for (int c = 0; c < 3; c++) {
                    const float m = d2[c] - 2.f * d1[c] + d0[c];
                    if (t == 0) acc += w_mom * fabsf(m);
                    const float sg = m > 0.f ? w_mom : (m < 0.f ? -w_mom : 0.f);
                    g[c] += t == 1 ? -2.f * sg : sg;
                }
            }
            gx += g[0]; gy += g[1]; gz += g[2];
        }
        float *o = grad + ((size_t)t * V + v) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    const unsigned nblocks = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
        partial[me] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == nblocks - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    float s = 0.f;
    for (unsigned j = threadIdx.x; j < nblocks; j += 256) s += __builtin_nontemporal_load(partial + j);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        *loss = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        *ticket = 0u;
    }
}

template <int T>
int launch_fwd(hipStream_t s, int R, const float *W, const float *b, const float *h, float *y, const float *add) {
    // (forward: one pass of four rows per wave when the rows allow it -- every load of the launch in flight at once; 512 workgroups
    //  walking ~4 dependent load -> reduce rounds each measured 25 us for 31 MB)
    const int fwd_blocks = std::max(SIM_BLOCKS, std::min(2048, (R + 15) / 16));
    k_rows_dot_fwd<T><<<fwd_blocks, 256, 0, s>>>(R, (const float4 *)W, b, (const float4 *)h, y, add);
    LAUNCH_CHECK();
    return 0;
}
template <int T>
int launch_bwd(hipStream_t s, int R, const float *W, const float *h, const float *dy, float *dW, float *db, float *dh, void *scratch) {
    k_rows_dot_bwd<T><<<SIM_BLOCKS, 256, 0, s>>>(R, (const float4 *)W, (const float4 *)h, dy, (float4 *)dW, db, (float4 *)scratch);
    LAUNCH_CHECK();
    k_rows_dot_dh<<<cdiv(T * SIM_K, 64), 1024, 0, s>>>(T * SIM_K, SIM_BLOCKS, (const float *)scratch, dh);
    LAUNCH_CHECK();
    return 0;
}
// ---- the two hidden layers of the simulator MLP for the T time rows of a step, ONE launch each way
// (meshnet_network.py:337-338,364-366: relu(Linear(13, 256)) -> relu(Linear(256, 256)) on the sinusoidal code of one time value per
// camera).  As torch ops the step pays ~20 launches for them (two [T,*] GEMMs and their four backward GEMMs at M = T = 3, bias sums,
// ReLU masks): ~100 us of GPU time at ~5 us per launch floor and more on the host, for 70 k multiply-adds.  One 256-thread
// workgroup: thread j owns hidden unit j.
//   forward : h1[t][j] = relu(b1[j] + sum_c W1[j][c] e[t][c]),  h2[t][j] = relu(b2[j] + sum_k W2[j][k] h1[t][k])
//   backward: dz2 = dh2 * (h2 > 0); db2 = sum_t dz2; dW2[j][k] = sum_t dz2[t][j] h1[t][k]; dh1[t][k] = sum_j W2[j][k] dz2[t][j];
//             dz1 = dh1 * (h1 > 0); db1 = sum_t dz1; dW1[k][c] = sum_t dz1[t][k] e[t][c]          (fixed summation order)
constexpr int SIM_H = 256, SIM_K0MAX = 16, SIM_FWD_WGS = 16;
template <int T>
__global__ __launch_bounds__(SIM_H) void k_sim_hidden_fwd(int K0, const float *__restrict__ e, const float *__restrict__ W1,
                                                          const float *__restrict__ b1, const float *__restrict__ W2,
                                                          const float *__restrict__ b2, float *__restrict__ h1, float *__restrict__ h2) {
    // SIM_FWD_WGS workgroups: every one forms the whole first layer in LDS (13 x 256 multiply-adds, redundantly) and then ITS 16 units of
    // the second: sixteen lanes per unit read the unit's W2 row as four 16-byte pieces each (coalesced over the row) and meet in a DPP row
    // sum.  (One workgroup with a thread per unit walked 64 dependent 16-byte loads at a 1 KB stride: 12-13 us.)  The order of every sum is
    // the same for every T: a T = 1 forward gives the bits of the T = 3 one.
    __shared__ float s_h1[T][SIM_H];
    const int j = threadIdx.x, g = blockIdx.x;
    float a[T];
#pragma unroll
    for (int t = 0; t < T; t++) a[t] = b1[j];
    for (int c = 0; c < K0; c++) {
        const float w = W1[j * K0 + c];
#pragma unroll
        for (int t = 0; t < T; t++) a[t] = __fmaf_rn(w, e[t * K0 + c], a[t]);
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        a[t] = fmaxf(a[t], 0.f);
        s_h1[t][j] = a[t];
        if (j / (SIM_H / SIM_FWD_WGS) == g) h1[t * SIM_H + j] = a[t];
    }
    __syncthreads();
    const int unit = g * (SIM_H / SIM_FWD_WGS) + (j >> 4), l = j & 15;
    float o[T];
#pragma unroll
    for (int t = 0; t < T; t++) o[t] = 0.f;
    const float4 *row = reinterpret_cast<const float4 *>(W2 + (size_t)unit * SIM_H);
    float4 w[4];
#pragma unroll
    for (int m = 0; m < 4; m++) w[m] = row[l + 16 * m];
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const float4 x = *reinterpret_cast<const float4 *>(&s_h1[t][4 * (l + 16 * m)]);
            o[t] = __fmaf_rn(w[m].w, x.w, __fmaf_rn(w[m].z, x.z, __fmaf_rn(w[m].y, x.y, __fmaf_rn(w[m].x, x.x, o[t]))));
        }
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        float v = o[t];
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));   // row_mirror
        if (l == 0) h2[t * SIM_H + unit] = fmaxf(v + b2[unit], 0.f);
    }
}

constexpr int SIM_BWD_WGS = 16;           // workgroups of the backward: 16 rows of W2 / dW2 each
template <int T>
__global__ __launch_bounds__(SIM_H) void k_sim_hidden_bwd(int K0, const float *__restrict__ e, const float *__restrict__ W2,
                                                          const float *__restrict__ h1, const float *__restrict__ h2,
                                                          const float *__restrict__ dh2, float *__restrict__ dW1, float *__restrict__ db1,
                                                          float *__restrict__ dW2, float *__restrict__ db2,
                                                          float *__restrict__ part, unsigned *__restrict__ ticket) {
    // workgroup g owns rows j = 16 g .. 16 g + 15 of W2: their dW2 rows and db2 entries are complete here; its share of
    // dh1[t][k] = sum_j W2[j][k] dz2[t][j] goes to part[g][t][k], and the workgroup that finishes LAST (ticket) adds the 16 shares in index
    // order (deterministic) and does the first layer (one workgroup took 34 us for the whole thing: a 256-iteration chain of row loads)
    __shared__ float s_dz2[T][SIM_H / SIM_BWD_WGS];
    __shared__ bool s_last;
    const int k = threadIdx.x, g = blockIdx.x;
    constexpr int RW = SIM_H / SIM_BWD_WGS;
    float x1[T], d1[T];
#pragma unroll
    for (int t = 0; t < T; t++) { x1[t] = h1[t * SIM_H + k]; d1[t] = 0.f; }
    if (k < RW) {
        const int j = g * RW + k;
        float sb = 0.f;
#pragma unroll
        for (int t = 0; t < T; t++) {
            const float dz = h2[t * SIM_H + j] > 0.f ? dh2[t * SIM_H + j] : 0.f;
            s_dz2[t][k] = dz;
            sb += dz;
        }
        db2[j] = sb;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < RW; jj++) {
        const int j = g * RW + jj;
        const float w = W2[(size_t)j * SIM_H + k];
        float gsum = 0.f;
#pragma unroll
        for (int t = 0; t < T; t++) {
            const float dz = s_dz2[t][jj];
            gsum = __fmaf_rn(dz, x1[t], gsum);
            d1[t] = __fmaf_rn(w, dz, d1[t]);
        }
        dW2[(size_t)j * SIM_H + k] = gsum;
    }
#pragma unroll
    for (int t = 0; t < T; t++) part[((size_t)g * T + t) * SIM_H + k] = d1[t];
    __threadfence();
    __syncthreads();
    if (k == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    float sb1 = 0.f;
#pragma unroll
    for (int t = 0; t < T; t++) {
        float a = 0.f;
        for (int gg = 0; gg < SIM_BWD_WGS; gg++) a += __builtin_nontemporal_load(part + ((size_t)gg * T + t) * SIM_H + k);
        d1[t] = x1[t] > 0.f ? a : 0.f;
        sb1 += d1[t];
    }
    db1[k] = sb1;
    for (int c = 0; c < K0; c++) {
        float gsum = 0.f;
#pragma unroll
        for (int t = 0; t < T; t++) gsum = __fmaf_rn(d1[t], e[t * K0 + c], gsum);
        dW1[k * K0 + c] = gsum;
    }
    if (k == 0) *ticket = 0u;          // (left at zero for the next call on this scratch)
}

}  // namespace

extern "C" {

size_t csplat_rows_dot_scratch_bytes(int T) { return align256((size_t)SIM_BLOCKS * (T > 0 ? T : 1) * SIM_K * sizeof(float)); }

int csplat_rows_dot_fwd(void *stream, int T, int R, int K, const float *W, const float *b, const float *h, float *y, const float *add) {
    CSPLAT_REQUIRE(K == SIM_K, "csplat_rows_dot_fwd: K must be 256");
    CSPLAT_REQUIRE(T >= 0 && T <= SIM_TMAX && R >= 0, "csplat_rows_dot_fwd: T must be 0..8, R >= 0");
    if (T == 0 || R == 0) return 0;
    CSPLAT_REQUIRE(W && b && h && y, "csplat_rows_dot_fwd: NULL");
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 1: return launch_fwd<1>(s, R, W, b, h, y, add);
        case 2: return launch_fwd<2>(s, R, W, b, h, y, add);
        case 3: return launch_fwd<3>(s, R, W, b, h, y, add);
        case 4: return launch_fwd<4>(s, R, W, b, h, y, add);
        case 5: return launch_fwd<5>(s, R, W, b, h, y, add);
        case 6: return launch_fwd<6>(s, R, W, b, h, y, add);
        case 7: return launch_fwd<7>(s, R, W, b, h, y, add);
        default: return launch_fwd<8>(s, R, W, b, h, y, add);
    }
}

int csplat_rows_dot_bwd(void *stream, int T, int R, int K, const float *W, const float *h, const float *dy, float *dW, float *db,
                        float *dh, void *scratch) {
    CSPLAT_REQUIRE(K == SIM_K, "csplat_rows_dot_bwd: K must be 256");
    CSPLAT_REQUIRE(T >= 0 && T <= SIM_TMAX && R >= 0, "csplat_rows_dot_bwd: T must be 0..8, R >= 0");
    if (T == 0) return 0;
    CSPLAT_REQUIRE(h && dh && scratch && (R == 0 || (W && dy && dW && db)), "csplat_rows_dot_bwd: NULL");
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 1: return launch_bwd<1>(s, R, W, h, dy, dW, db, dh, scratch);
        case 2: return launch_bwd<2>(s, R, W, h, dy, dW, db, dh, scratch);
        case 3: return launch_bwd<3>(s, R, W, h, dy, dW, db, dh, scratch);
        case 4: return launch_bwd<4>(s, R, W, h, dy, dW, db, dh, scratch);
        case 5: return launch_bwd<5>(s, R, W, h, dy, dW, db, dh, scratch);
        case 6: return launch_bwd<6>(s, R, W, h, dy, dW, db, dh, scratch);
        case 7: return launch_bwd<7>(s, R, W, h, dy, dW, db, dh, scratch);
        default: return launch_bwd<8>(s, R, W, h, dy, dW, db, dh, scratch);
    }
}

size_t csplat_cloth_regs_scratch_bytes(int T, int V, int64_t E) {
    const int64_t items = (int64_t)V + (int64_t)(T > 0 ? T : 0) * (E > 0 ? E : 0);
    const int64_t blocks = cdiv(items > 0 ? items : 1, 256) + (int64_t)(T > 0 ? T : 0) * cdiv(V > 0 ? V : 1, 256);
    return align256((size_t)(blocks + 1) * sizeof(float)) + 256;
}

int csplat_cloth_regs(void *stream, int T, int V, int64_t E, const float *D, const int64_t *edge_index, const float *rest_len,
                      float lambda_deform, float lambda_rigid, float lambda_momentum, float *loss, float *grad, void *scratch,
                      const int *dst_rowptr, const int *dst_perm, const int *src_rowptr, const int *src_perm) {
    CSPLAT_REQUIRE(T >= 0 && V >= 0 && E >= 0, "csplat_cloth_regs: bad sizes");
    CSPLAT_REQUIRE(loss && scratch, "csplat_cloth_regs: NULL loss / scratch");
    hipStream_t s = (hipStream_t)stream;
    const bool node_terms = T >= 3 && V > 0 && (lambda_deform != 0.f || lambda_momentum != 0.f);
    const bool edge_terms = T > 0 && E > 0 && V > 0 && lambda_rigid != 0.f;
    const bool csr = dst_rowptr != nullptr;
    CSPLAT_REQUIRE(!csr || (dst_perm && src_rowptr && src_perm), "csplat_cloth_regs: incomplete CSR");
    CSPLAT_REQUIRE(T < 65536, "csplat_cloth_regs: T too large");
    if (T > 0 && V > 0) {
        CSPLAT_REQUIRE(D && grad, "csplat_cloth_regs: NULL D / grad");
        if (!csr || !(node_terms || edge_terms)) HIP_TRY(hipMemsetAsync(grad, 0, (size_t)T * V * 3 * sizeof(float), s));
    }
    if (!node_terms && !edge_terms) {
        HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), s));
        return 0;
    }
    CSPLAT_REQUIRE(!edge_terms || (edge_index && rest_len), "csplat_cloth_regs: NULL edge arrays");
    const int64_t items = (int64_t)V + (edge_terms ? (int64_t)T * E : 0);
    const int blocks = cdiv(items, 256);
    float *partial = (float *)scratch;
    // the ticket word sits behind all partials, in the last 256 bytes: ZERO on entry (the caller zeroes the scratch once), left at zero
    unsigned int *ticket = (unsigned int *)((char *)scratch + csplat_cloth_regs_scratch_bytes(T, V, E) - 256);
    if (csr) {
        k_cloth_regs_csr<<<dim3(cdiv(V, 256), T), 256, 0, s>>>(
            T, V, (long long)E, D, edge_index, rest_len, dst_rowptr, dst_perm, src_rowptr, src_perm,
            node_terms && lambda_deform != 0.f ? 0.5f * lambda_deform / (float)V : 0.f,
            edge_terms ? lambda_rigid / ((float)T * (float)E) : 0.f,
            node_terms && lambda_momentum != 0.f ? lambda_momentum / (float)V : 0.f, grad, partial, ticket, loss);
        LAUNCH_CHECK();
        return 0;
    }
    k_cloth_regs<<<blocks, 256, 0, s>>>(T, V, (long long)E, D, edge_index, rest_len,
                                        node_terms && lambda_deform != 0.f ? 0.5f * lambda_deform / (float)V : 0.f,
                                        edge_terms ? lambda_rigid / ((float)T * (float)E) : 0.f,
                                        node_terms && lambda_momentum != 0.f ? lambda_momentum / (float)V : 0.f, grad, partial, ticket,
                                        loss);
    LAUNCH_CHECK();
    return 0;
}

int csplat_sim_hidden_fwd(void *stream, int T, int K0, const float *e, const float *W1, const float *b1, const float *W2, const float *b2,
                          float *h1, float *h2) {
    CSPLAT_REQUIRE(T >= 1 && T <= SIM_TMAX && K0 >= 1 && K0 <= SIM_K0MAX, "csplat_sim_hidden_fwd: 1 <= T <= 8 time rows, 1 <= K0 <= 16 inputs");
    CSPLAT_REQUIRE(e && W1 && b1 && W2 && b2 && h1 && h2 && ((uintptr_t)W2 & 15u) == 0, "csplat_sim_hidden_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
#define CSPLAT_SIMH(TT) case TT: k_sim_hidden_fwd<TT><<<SIM_FWD_WGS, SIM_H, 0, s>>>(K0, e, W1, b1, W2, b2, h1, h2); break;
    switch (T) { CSPLAT_SIMH(1) CSPLAT_SIMH(2) CSPLAT_SIMH(3) CSPLAT_SIMH(4) CSPLAT_SIMH(5) CSPLAT_SIMH(6) CSPLAT_SIMH(7) CSPLAT_SIMH(8) }
#undef CSPLAT_SIMH
    LAUNCH_CHECK();
    return 0;
}
size_t csplat_sim_hidden_scratch_bytes(int T) { return ((size_t)SIM_BWD_WGS * (size_t)(T > 0 ? T : 1) * SIM_H + 64) * 4; }
int csplat_sim_hidden_bwd(void *stream, int T, int K0, const float *e, const float *W2, const float *h1, const float *h2, const float *dh2,
                          float *dW1, float *db1, float *dW2, float *db2, void *scratch) {
    CSPLAT_REQUIRE(T >= 1 && T <= SIM_TMAX && K0 >= 1 && K0 <= SIM_K0MAX, "csplat_sim_hidden_bwd: 1 <= T <= 8 time rows, 1 <= K0 <= 16 inputs");
    CSPLAT_REQUIRE(e && W2 && h1 && h2 && dh2 && dW1 && db1 && dW2 && db2 && scratch, "csplat_sim_hidden_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    unsigned *ticket = (unsigned *)scratch;                 // first word: zero on entry, left at zero
    float *part = (float *)scratch + 64;
#define CSPLAT_SIMH(TT) case TT: k_sim_hidden_bwd<TT><<<SIM_BWD_WGS, SIM_H, 0, s>>>(K0, e, W2, h1, h2, dh2, dW1, db1, dW2, db2, part, ticket); break;
    switch (T) { CSPLAT_SIMH(1) CSPLAT_SIMH(2) CSPLAT_SIMH(3) CSPLAT_SIMH(4) CSPLAT_SIMH(5) CSPLAT_SIMH(6) CSPLAT_SIMH(7) CSPLAT_SIMH(8) }
#undef CSPLAT_SIMH
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
