// csplat_gemm.hip -- the 128-wide Linear layers of the MeshNet MLPs for INFERENCE (rollout,
// BASELINE configs[3]): out[M][128] = act(A[M][128] @ W[128][128]^T + bias), fp32 in, fp32 accumulate, exact-fp32 MFMA.
//
// The reference runs these through cuBLAS sgemm (meshnet/graph_network.py:198,221 via nn.Linear); on this stack the same
// call (rocBLAS / hipBLASLt, M = 300,000 rows, N = K = 128) reaches ~34 TFLOP/s, i.e. ~290 us per layer, and the 45 such
// layers of a rollout step are 85 % of its time.  The shape is fixed and skinny, so a dedicated kernel can sit on both
// roofs at once: 9.8 GFLOP at the 155 TFLOP/s fp32-MFMA rate is 63 us, 2 x 154 MB at ~5 TB/s is 62 us.
//
// Design (gfx950): persistent workgroups of 4 wavefronts; W^T is staged ONCE per workgroup in LDS (k-major, 64.5 KB);
// a wave owns 32 output rows x 128 columns: 4 accumulator tiles of v_mfma_f32_32x32x2_f32 (64 VGPRs).  The contraction
// index is permuted so that lane-half h of a wave takes k = 64h + s at step s: each lane then needs ONE contiguous
// 256-byte run of its A row (16 x global_load_dwordx4, no redundancy between the halves, no LDS for A), and the B operand
// of step s is one conflict-free ds_read_b32 per column tile (32 consecutive columns at row k of W^T).  Bias + ReLU are
// applied to the accumulators; each (register, lane-half) stores 32 consecutive floats of one output row.
#include "csplat_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// sum over the 32 lanes of a half wave (lanes 0..31 / 32..63), result in every lane of the half: four DPP adds inside each
// 16-lane row, then the gfx950 row swap folds rows 0|1 and 2|3 (cheaper than five ds_bpermute round trips)
template <int CTRL>
__device__ __forceinline__ float l128_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_sum(float v) {
    v = l128_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = l128_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = l128_dpp_add<0x141>(v);   // row_half_mirror
    v = l128_dpp_add<0x140>(v);   // row_mirror
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}
constexpr int GK = 128, GN = 128, WT_STRIDE = 129;   // W^T rows padded: conflict-free transposed staging

// Epilogue options (all fused into the accumulator registers, no extra HBM pass):
//   alpha     : out = alpha * (A @ W^T) + bias                (the 2^l edge-feature scale of graph_network.py:222)
//   GATHER    : + ga[ia[row]][col] + gb[ib[row]][col]          (the x_i / x_j column blocks of the split first Linear)
//   relu      : max(., 0)
//   LN        : LayerNorm over the 128 columns of each row (biased variance, eps), then * gamma + beta
//   ADD       : + add_pre[row][col] before the ReLU and/or + add_post[row][col] after the LayerNorm (row-aligned [M][128]
//               operands: the second half of a split first Linear, the residual connection).  Loaded inline between the
//               stores, i.e. off the tuned path: meant for the node-level calls (M = N nodes), not the edge-level ones.
//   B3        : the product itself through bf16 MFMAs instead of fp32 MFMAs: both operands are cut into three bf16 pieces
//               (x = x1 + x2 + x3 exactly to 24 bits: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)) and the six
//               partial products that matter (x1w1, x1w2, x2w1, x1w3, x3w1, x2w2; the dropped ones are < 2^-24 relative) are
//               accumulated in fp32 by v_mfma_f32_32x32x16_bf16, which runs at 16x the fp32 MFMA rate on gfx950 (measured
//               2.4 PFLOP/s vs 155 TFLOP/s): 6/16 of the MFMA time for fp32-level accuracy (tested: 1e-5 vs fp64).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WB_STRIDE = 136;        // bf16 elements per W row in LDS (272 B: conflict-free 16-byte reads at row stride)
constexpr size_t L128_LDS_F32 = (size_t)GK * WT_STRIDE * 4, L128_LDS_B3 = (size_t)3 * GN * WB_STRIDE * 2;

template <bool GATHER, bool LN, bool ADD, bool B3>
__global__ __launch_bounds__(B3 ? 512 : 256, 2) void k_linear128(int64_t M, const float *A, const float *__restrict__ W,
                                                    const float *__restrict__ bias, float alpha, int relu,
                                                    const float *__restrict__ ga, const int64_t *__restrict__ ia,
                                                    const float *__restrict__ gb, const int64_t *__restrict__ ib,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                    const float *add_pre, const float *add_post, float *out,
                                                    int ldw, int wt, const float *mask, float2 *ln_stats) {
    // W is read as W[j][k] = W[j * ldw + k] (wt = 0: a torch Linear.weight, or a column slice of a wider one) or as
    // W[k * ldw + j] (wt = 1: the TRANSPOSE of such a matrix -- the input-gradient product -- without a transposed copy)
    extern __shared__ float s_wt[];   // fp32 path: [GK][WT_STRIDE], s_wt[k * WT_STRIDE + j] = W[j][k];  B3: bf16 [3][GN][WB_STRIDE]
    constexpr int NW = B3 ? 8 : 4;    // wavefronts per workgroup
    // TR: the variant with row-aligned [M][128] epilogue operands (the autograd path's mask / running-sum addends) forms the product
    // transposed -- MFMA A operand = weight, B operand = rows -- so that a lane ends up with 64 outputs of ITS OWN row in groups of
    // four consecutive columns (register r of tile c <-> column 32 c + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): the operands are then
    // 16 float4 loads and the result 16 float4 stores per lane instead of 64 scalar ones each
    constexpr bool TR = B3 && ADD && !LN && !GATHER;
    if (B3) {
        __bf16 *wb = reinterpret_cast<__bf16 *>(s_wt);
        // element t of the 128 x 128 matrix AS STORED sits at (t >> 7) * ldw + (t & 127) whichever way it is read (consecutive
        // threads read consecutive addresses); only its place in LDS differs.  All 32 loads of a thread are in flight together.
        constexpr int PER = GN * GK / (NW * 64);
        float xs[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int t = threadIdx.x + i * NW * 64;
            xs[i] = W[(size_t)(t >> 7) * ldw + (t & 127)];
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int t = threadIdx.x + i * NW * 64;
            const int j = wt ? t & 127 : t >> 7, k = wt ? t >> 7 : t & 127;
            const float x = xs[i];
            const __bf16 p1 = (__bf16)x;
            const float r1 = x - (float)p1;
            const __bf16 p2 = (__bf16)r1;
            const __bf16 p3 = (__bf16)(r1 - (float)p2);
            wb[(size_t)j * WB_STRIDE + k] = p1;
            wb[(size_t)(GN + j) * WB_STRIDE + k] = p2;
            wb[(size_t)(2 * GN + j) * WB_STRIDE + k] = p3;
        }
    } else {   // all 16 float4 loads of a thread in flight at once; W[j][4i..4i+3] -> rows 4i..4i+3 of W^T
        float4 wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int t = threadIdx.x + 256 * i, j = t >> 5, k = (t & 31) * 4;
            if (!wt && !(ldw & 3)) wv[i] = *reinterpret_cast<const float4 *>(W + (size_t)j * ldw + k);
            else if (!wt) wv[i] = make_float4(W[(size_t)j * ldw + k], W[(size_t)j * ldw + k + 1], W[(size_t)j * ldw + k + 2], W[(size_t)j * ldw + k + 3]);
            else wv[i] = make_float4(W[(size_t)k * ldw + j], W[(size_t)(k + 1) * ldw + j], W[(size_t)(k + 2) * ldw + j], W[(size_t)(k + 3) * ldw + j]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int t = threadIdx.x + 256 * i, j = t >> 5, k = (t & 31) * 4;
            s_wt[(k + 0) * WT_STRIDE + j] = wv[i].x;
            s_wt[(k + 1) * WT_STRIDE + j] = wv[i].y;
            s_wt[(k + 2) * WT_STRIDE + j] = wv[i].z;
            s_wt[(k + 3) * WT_STRIDE + j] = wv[i].w;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int64_t ntile = (M + 31) / 32;
    float bcol[4], gcol[4], becol[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        bcol[c] = bias ? bias[32 * c + r32] : 0.f;
        gcol[c] = LN ? gamma[32 * c + r32] : 1.f;
        becol[c] = LN ? beta[32 * c + r32] : 0.f;
    }
    // Memory-op choreography.  On gfx9-family ISAs loads and stores share vmcnt and may retire out of order with respect
    // to each other, so a wait on ANY load while stores are in flight is a wait for those stores.  Each iteration
    // therefore (1) runs this tile's 256 MFMAs, re-filling each 16-byte chunk of the A-row registers with the NEXT tile's
    // data as soon as its last MFMA has consumed it (no second register set), (2) waits for those loads -- all but the
    // last few long since landed -- and only then (3) issues this tile's stores, which drain under the next tile's MFMAs.
    // Rows past M are clamped to M - 1 (loads stay unconditional; MFMA rows are independent, masked at the store).
    auto row_of = [&](int64_t tile) {
        const int64_t row = tile * 32 + r32;
        return row < M ? row : M - 1;
    };
    const int64_t tstride = (int64_t)gridDim.x * NW;
    float4 X[16];                                  // k = 64h .. 64h + 63 of this lane's row
    int ja = 0, jb = 0, jan = 0, jbn = 0;          // gather rows of tile row r32 (node ids: < 2^31)
    {
        const int64_t row = row_of((int64_t)blockIdx.x * NW + w);
        const float4 *ap = reinterpret_cast<const float4 *>(A + row * GK + 64 * h);
#pragma unroll
        for (int q = 0; q < 16; q++) X[q] = ap[q];
        if (GATHER) { ja = (int)ia[row]; jb = (int)ib[row]; }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing from the prologue is pending at loop entry
    const float *wrow = s_wt + (64 * h) * WT_STRIDE + r32;
    for (int64_t tile = (int64_t)blockIdx.x * NW + w; tile < ntile; tile += tstride) {
        const int64_t nrow = row_of(tile + tstride);
        const float4 *apn = reinterpret_cast<const float4 *>(A + nrow * GK + 64 * h);
        if (GATHER) { jan = (int)ia[nrow]; jbn = (int)ib[nrow]; }
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
        if constexpr (B3) {
            // step s covers k = 64h + 8s .. + 7 of this lane's row (chunks X[2s], X[2s + 1]); the B operand of (split p,
            // column tile c, step s) is the 16 contiguous bytes wb[p][32c + r32][64h + 8s ..]: W is staged untransposed.
            // The next tile's rows are fetched into a second register set at the top of the phase (the bf16 phase is too
            // short to hide the last in-place re-fills).
            float4 Y[16];
#pragma unroll
            for (int q = 0; q < 16; q++) Y[q] = apn[q];
            const __bf16 *wlane = reinterpret_cast<const __bf16 *>(s_wt) + (size_t)r32 * WB_STRIDE + 64 * h;
            auto ldw = [&](int p, int c, int st) {
                return *reinterpret_cast<const bf16x8 *>(wlane + ((size_t)p * GN + 32 * c) * WB_STRIDE + 8 * st);
            };
            bf16x8 wc[3], wn[3];
#pragma unroll
            for (int p = 0; p < 3; p++) wc[p] = ldw(p, 0, 0);
#pragma unroll
            for (int st = 0; st < 8; st++) {
                const float x8[8] = {X[2 * st].x, X[2 * st].y, X[2 * st].z, X[2 * st].w,
                                     X[2 * st + 1].x, X[2 * st + 1].y, X[2 * st + 1].z, X[2 * st + 1].w};
                bf16x8 a1, a2, a3;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    a1[j] = (__bf16)x8[j];
                    const float r1 = x8[j] - (float)a1[j];
                    a2[j] = (__bf16)r1;
                    a3[j] = (__bf16)(r1 - (float)a2[j]);
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int cn = (c + 1) & 3, sn = c == 3 ? st + 1 : st;
                    if (sn < 8) {
#pragma unroll
                        for (int p = 0; p < 3; p++) wn[p] = ldw(p, cn, sn);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // next operands requested before this group's 6 MFMAs (192 cycles)
                    if constexpr (TR) {   // weights as the A operand: the accumulators hold the TRANSPOSED tile (see the epilogue)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a3, acc[c], 0, 0, 0);
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2], a1, acc[c], 0, 0, 0);
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[1], a2, acc[c], 0, 0, 0);
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a2, acc[c], 0, 0, 0);
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[1], a1, acc[c], 0, 0, 0);
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a1, acc[c], 0, 0, 0);
                    } else {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, wc[0], acc[c], 0, 0, 0);   // small terms first
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wc[2], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wc[1], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wc[0], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wc[1], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wc[0], acc[c], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 3; p++) wc[p] = wn[p];
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile's rows (requested a whole phase ago), before any store
#pragma unroll
            for (int q = 0; q < 16; q++) X[q] = Y[q];
        } else {
            // B operand: step s (this lane contributes k = 64h + s) and column tile c read s_wt[(64h + s)][32c + r32].
            // Explicitly double-buffered in groups of 2 steps: the 8 ds_reads of group g + 1 are issued before the 8 MFMAs
            // (512 cycles) of group g, so the MFMA pipe never waits on LDS latency; sched_barrier pins the order.
            float bc[8], bn[8];
    #pragma unroll
            for (int u = 0; u < 2; u++)
    #pragma unroll
                for (int c = 0; c < 4; c++) bc[4 * u + c] = wrow[u * WT_STRIDE + 32 * c];
    #pragma unroll
            for (int g = 0; g < 32; g++) {
                if (g < 31) {
    #pragma unroll
                    for (int u = 0; u < 2; u++)
    #pragma unroll
                        for (int c = 0; c < 4; c++) bn[4 * u + c] = wrow[(2 * (g + 1) + u) * WT_STRIDE + 32 * c];
                }
                __builtin_amdgcn_sched_barrier(0);     // reads first: they complete under this group's 512 MFMA cycles
                const float4 xq = X[g >> 1];
                const float av[2] = {(g & 1) ? xq.z : xq.x, (g & 1) ? xq.w : xq.y};
    #pragma unroll
                for (int u = 0; u < 2; u++)
    #pragma unroll
                    for (int c = 0; c < 4; c++)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bc[4 * u + c], acc[c], 0, 0, 0);
                if (g & 1) X[g >> 1] = apn[g >> 1];   // chunk consumed: fetch the next tile's into the same registers
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int i = 0; i < 8; i++) bc[i] = bn[i];
            }
        }
        if constexpr (TR) {
            const int64_t orow = tile * 32 + r32;
            const bool live = orow < M;
            const int64_t arow = live ? orow : M - 1;
            auto row4 = [&](const float *p, float4 (&v)[16]) __attribute__((always_inline)) {   // the lane's 64 columns of its row
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int q = 0; q < 4; q++) v[4 * c + q] = *reinterpret_cast<const float4 *>(p + arow * GN + 32 * c + 8 * q + 4 * h);
            };
            auto each = [&](auto fn) __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[c][4 * q + j] = fn(acc[c][4 * q + j], 4 * c + q, j, 32 * c + 8 * q + 4 * h + j);
            };
            auto comp = [](const float4 &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };
            if (bias) {
                float4 t[16];
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int q = 0; q < 4; q++) t[4 * c + q] = *reinterpret_cast<const float4 *>(bias + 32 * c + 8 * q + 4 * h);
                each([&](float a, int i, int j, int) { return alpha * a + comp(t[i], j); });
            } else {
                each([&](float a, int, int, int) { return alpha * a; });
            }
            if (add_pre) { float4 t[16]; row4(add_pre, t); each([&](float a, int i, int j, int) { return a + comp(t[i], j); }); }
            if (relu) each([&](float a, int, int, int) { return relu_keep_nan(a); });
            if (add_post) { float4 t[16]; row4(add_post, t); each([&](float a, int i, int j, int) { return a + comp(t[i], j); }); }
            if (mask) { float4 t[16]; row4(mask, t); each([&](float a, int i, int j, int) { return comp(t[i], j) > 0.f ? a : 0.f; }); }
            if (live) {
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        *reinterpret_cast<float4 *>(out + orow * GN + 32 * c + 8 * q + 4 * h) =
                            make_float4(acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]);
            }
            continue;
        }
        // C/D layout of 32x32 tiles: column = lane & 31, tile row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        float o[16][4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[r][c] = alpha * acc[c][r] + bcol[c];
        if (GATHER) {
            // The gather rows of the tile go through a 256-byte wave-private LDS strip (DS ops of one wave execute in
            // order: no barrier), so each row's index is one broadcast ds_read instead of a cross-lane shuffle.
            int *strip = reinterpret_cast<int *>(reinterpret_cast<char *>(s_wt) + (B3 ? L128_LDS_B3 : L128_LDS_F32)) + w * 64;
            if (h == 0) { strip[r32] = ja; strip[32 + r32] = jb; }
            // four batches of 4 rows: the 32 gather loads of a batch are in flight together (no stores pending here)
#pragma unroll
            for (int b4 = 0; b4 < 4; b4++) {
                float g[4][4][2];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int r = 4 * b4 + rr, trow = (r & 3) + 8 * (r >> 2);
                    const int64_t ra = strip[trow + 4 * h], rb = strip[32 + trow + 4 * h];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        g[rr][c][0] = ga[ra * GN + 32 * c + r32];
                        g[rr][c][1] = gb[rb * GN + 32 * c + r32];
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 4; rr++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        o[4 * b4 + rr][c] += g[rr][c][0] + g[rr][c][1];
                        asm volatile("" : "+v"(o[4 * b4 + rr][c]));   // pin the add here: keeps the gathered values short-lived
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // The wait on this phase's loads, made explicit so that it precedes the stores.  The 6 youngest loads (chunks
        // 10..15 of the next tile, issued in the last third of the MFMA phase) may stay in flight: they are not needed
        // before step 40 of the next tile, by which time these stores have drained.  (A second register set filled at
        // the top of the phase was measured: same time, 60 more VGPRs.)
        if (!B3) __builtin_amdgcn_s_waitcnt(0x0F76);   // vmcnt(6)
        ja = jan; jb = jbn;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = o[r][c];
            const int64_t arow = orow < M ? orow : M - 1;
            if (ADD && add_pre) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] += add_pre[arow * GN + 32 * c + r32];
            }
            if (relu) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = relu_keep_nan(v[c]);
            }
            if (LN) {
                float sum = (v[0] + v[1]) + (v[2] + v[3]);
                sum = half_sum(sum);
                const float mean = sum * (1.f / GN);
                float d[4], sq = 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) { d[c] = v[c] - mean; sq += d[c] * d[c]; }
                sq = half_sum(sq);
                const float rstd = rsqrtf(sq * (1.f / GN) + eps);
                if (ln_stats && r32 == 0 && orow < M) ln_stats[orow] = make_float2(mean, rstd);
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = d[c] * rstd * gcol[c] + becol[c];
            }
            if (ADD && add_post) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] += add_post[arow * GN + 32 * c + r32];
            }
            if (ADD && mask) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = mask[arow * GN + 32 * c + r32] > 0.f ? v[c] : 0.f;
            }
            if (orow < M) {
#pragma unroll
                for (int c = 0; c < 4; c++) out[orow * GN + 32 * c + r32] = v[c];
            }
        }
    }
}


// Node-level variant (M = N nodes, ~10^4 rows): with one 32-row tile per wave the persistent kernel above fills a third
// of the chip and pays a 64 KB weight staging per workgroup.  Here a workgroup owns ONE tile and its 4 waves split the 128
// output columns: each wave stages only its 16 KB of W^T, runs 64 MFMAs, and LayerNorm statistics cross the waves through
// LDS.  Same arithmetic, same epilogue options minus the gathers.
constexpr int SW_STRIDE = 33;
template <bool LN>
__global__ __launch_bounds__(256) void k_linear128_rows32(int64_t M, const float *A, const float *__restrict__ W,
                                                           const float *__restrict__ bias, float alpha, int relu,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                           const float *add_pre, const float *add_post, float *out,
                                                           int ldw, int wt, const float *mask, float2 *ln_stats) {
    extern __shared__ float s_dyn[];
    float (*s_w)[GK * SW_STRIDE] = reinterpret_cast<float (*)[GK * SW_STRIDE]>(s_dyn);   // per wave: s_w[w][k * 33 + jj] = W[32w + jj][k]
    float (*s_part)[32][4] = reinterpret_cast<float (*)[32][4]>(s_dyn + 4 * GK * SW_STRIDE);   // [pass][row][wave] partial sums
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    {
        float4 wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = lane + 64 * i, j = 32 * w + (idx >> 5), k = (idx & 31) * 4;
            if (!wt && !(ldw & 3)) wv[i] = *reinterpret_cast<const float4 *>(W + (size_t)j * ldw + k);
            else if (!wt) wv[i] = make_float4(W[(size_t)j * ldw + k], W[(size_t)j * ldw + k + 1], W[(size_t)j * ldw + k + 2], W[(size_t)j * ldw + k + 3]);
            else wv[i] = make_float4(W[(size_t)k * ldw + j], W[(size_t)(k + 1) * ldw + j], W[(size_t)(k + 2) * ldw + j], W[(size_t)(k + 3) * ldw + j]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = lane + 64 * i, jj = idx >> 5, k = (idx & 31) * 4;
            s_w[w][(k + 0) * SW_STRIDE + jj] = wv[i].x;
            s_w[w][(k + 1) * SW_STRIDE + jj] = wv[i].y;
            s_w[w][(k + 2) * SW_STRIDE + jj] = wv[i].z;
            s_w[w][(k + 3) * SW_STRIDE + jj] = wv[i].w;
        }
    }
    const int64_t tile = blockIdx.x;
    int64_t row = tile * 32 + r32;
    row = row < M ? row : M - 1;
    float4 X[16];
    const float4 *ap = reinterpret_cast<const float4 *>(A + row * GK + 64 * h);
#pragma unroll
    for (int q = 0; q < 16; q++) X[q] = ap[q];
    const int col = 32 * w + r32;
    const float bcol = bias ? bias[col] : 0.f, gcol = LN ? gamma[col] : 1.f, becol = LN ? beta[col] : 0.f;
    // when out aliases A, every wave must have read the tile's rows before any wave overwrites them
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const float *wrow = &s_w[w][(64 * h) * SW_STRIDE + r32];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const float av[4] = {X[q].x, X[q].y, X[q].z, X[q].w};
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wrow[(4 * q + u) * SW_STRIDE], acc, 0, 0, 0);
    }
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * h;
        int64_t orow = tile * 32 + trow;
        orow = orow < M ? orow : M - 1;
        v[r] = alpha * acc[r] + bcol;
        if (add_pre) v[r] += add_pre[orow * GN + col];
        if (relu) v[r] = relu_keep_nan(v[r]);
    }
    if (LN) {
        float mean[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float sum = v[r];
            sum = half_sum(sum);
            if (r32 == 0) s_part[0][(r & 3) + 8 * (r >> 2) + 4 * h][w] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float *p = s_part[0][(r & 3) + 8 * (r >> 2) + 4 * h];
            mean[r] = ((p[0] + p[1]) + (p[2] + p[3])) * (1.f / GN);
            const float d = v[r] - mean[r];
            float sq = d * d;
            sq = half_sum(sq);
            if (r32 == 0) s_part[1][(r & 3) + 8 * (r >> 2) + 4 * h][w] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float *p = s_part[1][(r & 3) + 8 * (r >> 2) + 4 * h];
            const float rstd = rsqrtf(((p[0] + p[1]) + (p[2] + p[3])) * (1.f / GN) + eps);
            const int64_t srow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ln_stats && w == 0 && r32 == 0 && srow < M) ln_stats[srow] = make_float2(mean[r], rstd);
            v[r] = (v[r] - mean[r]) * rstd * gcol + becol;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < M) {
            float o = v[r];
            if (add_post) o += add_post[orow * GN + col];
            if (mask) o = mask[orow * GN + col] > 0.f ? o : 0.f;
            out[orow * GN + col] = o;
        }
    }
}

// ---- the whole node update of an InteractionNetwork layer (graph_network.py:203-222) in ONE launch, plus the two
// node-level products the NEXT layer's edge kernel gathers (x_i / x_j column blocks of its first Linear):
//   h   = relu(agg @ Wa^T + x @ Wx^T + b0);  h = relu(h @ W2^T + b2);  x' = LN(h @ W3^T + b3) + x;
//   xa' = x' @ Wi^T;  xb' = x' @ Wj^T                                    (skipped when Wi == NULL)
// Six 32-row x 128 x 128 products per workgroup; a workgroup owns one 32-row tile, its 4 waves split the 128 output
// columns (each stages only its own 16 KB slice of the current weight, wave-private), activations pass from layer to
// layer through a 32 x 128 LDS tile.  Replaces six launches of k_linear128_rows32 (each ~10 us of fixed latency for
// 10^4 rows) per message-passing step.
constexpr int ACT_STRIDE = 132, NU_KH = 64;   // weights are staged half of K at a time: 52 KB of LDS, 3 workgroups per CU
__global__ __launch_bounds__(256) void k_node_update(int64_t N, const float *__restrict__ agg, const float *__restrict__ x,
                                                      const float *__restrict__ Wa, const float *__restrict__ Wx,
                                                      const float *__restrict__ b0, const float *__restrict__ W2,
                                                      const float *__restrict__ b2, const float *__restrict__ W3,
                                                      const float *__restrict__ b3, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float eps, const float *__restrict__ Wi,
                                                      const float *__restrict__ Wj, float *__restrict__ x_new,
                                                      float *__restrict__ xa, float *__restrict__ xb) {
    extern __shared__ float s_dyn[];
    float (*s_w)[NU_KH * SW_STRIDE] = reinterpret_cast<float (*)[NU_KH * SW_STRIDE]>(s_dyn);   // per wave: half a W^T slice [64][33]
    float (*s_act)[ACT_STRIDE] = reinterpret_cast<float (*)[ACT_STRIDE]>(s_dyn + 4 * NU_KH * SW_STRIDE);   // [32 rows][128 (+4)]
    float (*s_part)[32][4] = reinterpret_cast<float (*)[32][4]>(s_dyn + 4 * NU_KH * SW_STRIDE + 32 * ACT_STRIDE);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const int64_t tile = blockIdx.x;
    const int col = 32 * w + r32;
    int64_t row = tile * 32 + r32;
    row = row < N ? row : N - 1;

    // lane-half h of a wave contributes k = 64 kh + 32 h + (0..31) in K-half kh: registers X[8 kh + q] = k .. k + 3 at q
    // weight staging is software-pipelined: `prefetch` issues the global loads of the NEXT half slice into registers while
    // the MFMAs of the current one run; `commit` drops them into this wave's LDS slice (wave-private: no barrier)
    float4 wv[8];
    auto prefetch = [&](const float *W, int kh) {   // this wave's 32 output features, K-half kh
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = lane + 64 * i, jj = idx >> 4, k4 = (idx & 15) * 4;
            wv[i] = *reinterpret_cast<const float4 *>(W + (size_t)(32 * w + jj) * GK + 64 * kh + k4);
        }
    };
    auto commit = [&]() {                           // -> s_w[w][k'][jj]
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = lane + 64 * i, jj = idx >> 4, k4 = (idx & 15) * 4;
            s_w[w][(k4 + 0) * SW_STRIDE + jj] = wv[i].x;
            s_w[w][(k4 + 1) * SW_STRIDE + jj] = wv[i].y;
            s_w[w][(k4 + 2) * SW_STRIDE + jj] = wv[i].z;
            s_w[w][(k4 + 3) * SW_STRIDE + jj] = wv[i].w;
        }
    };
    auto mma_half = [&](const float4 (&X)[16], int kh, f32x16 acc) {
        const float *wrow = &s_w[w][(32 * h) * SW_STRIDE + r32];
        float bv[32];                         // all 32 B operands of the half requested up front: the MFMA chain then
#pragma unroll                                // only waits for the first one
        for (int t = 0; t < 32; t++) bv[t] = wrow[t * SW_STRIDE];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float4 xq = X[8 * kh + q];
            const float av[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
            for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[4 * q + u], acc, 0, 0, 0);
        }
        return acc;
    };
    // acc += X @ W^T[:, this wave's columns]; half 0 of W is already in `wv`, half 0 of `Wnext` is requested on the way out
    auto mma = [&](const float *W, const float *Wnext, const float4 (&X)[16], f32x16 acc) {
        commit();
        prefetch(W, 1);
        acc = mma_half(X, 0, acc);
        commit();
        if (Wnext) prefetch(Wnext, 0);
        return mma_half(X, 1, acc);
    };
    auto load_global = [&](const float *src, float4 (&X)[16]) {
#pragma unroll
        for (int q = 0; q < 16; q++)
            X[q] = *reinterpret_cast<const float4 *>(src + row * GK + 64 * (q >> 3) + 32 * h + 4 * (q & 7));
    };
    auto load_act = [&](float4 (&X)[16]) {
#pragma unroll
        for (int q = 0; q < 16; q++) X[q] = *reinterpret_cast<const float4 *>(&s_act[r32][64 * (q >> 3) + 32 * h + 4 * (q & 7)]);
    };
    auto zero = [&]() { f32x16 a; for (int r = 0; r < 16; r++) a[r] = 0.f; return a; };
    // C/D layout: column = lane & 31 (+ 32w), tile row = (reg & 3) + 8 * (reg >> 2) + 4h
    auto store_act = [&](const f32x16 &acc, const float *bias, bool relu) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = acc[r] + bv;
            s_act[(r & 3) + 8 * (r >> 2) + 4 * h][col] = relu ? relu_keep_nan(v) : v;
        }
    };
    float4 X[16], X2[16];
    // ---- layer 0: two products into one accumulator
    prefetch(Wa, 0);
    load_global(agg, X);
    load_global(x, X2);
    f32x16 acc = mma(Wa, Wx, X, zero());
    acc = mma(Wx, W2, X2, acc);
    store_act(acc, b0, true);
    __syncthreads();
    // ---- layer 1
    load_act(X);
    acc = mma(W2, W3, X, zero());
    __syncthreads();                          // every wave has read the tile before anyone overwrites it
    store_act(acc, b2, true);
    __syncthreads();
    // ---- layer 2 + LayerNorm + residual
    load_act(X);
    acc = mma(W3, Wi, X, zero());
    float v[16], mean[16];
    {
        const float bv = b3[col];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v[r] = acc[r] + bv;
            float sum = v[r];
            sum = half_sum(sum);
            if (r32 == 0) s_part[0][(r & 3) + 8 * (r >> 2) + 4 * h][w] = sum;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float *p = s_part[0][(r & 3) + 8 * (r >> 2) + 4 * h];
        mean[r] = ((p[0] + p[1]) + (p[2] + p[3])) * (1.f / GN);
        const float d = v[r] - mean[r];
        float sq = d * d;
        sq = half_sum(sq);
        if (r32 == 0) s_part[1][(r & 3) + 8 * (r >> 2) + 4 * h][w] = sq;
    }
    __syncthreads();                          // (also: every wave is done reading s_act)
    {
        const float gcol = gamma[col], becol = beta[col];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float *p = s_part[1][trow];
            const float rstd = rsqrtf(((p[0] + p[1]) + (p[2] + p[3])) * (1.f / GN) + eps);
            const int64_t orow = tile * 32 + trow;
            const int64_t crow = orow < N ? orow : N - 1;
            const float o = (v[r] - mean[r]) * rstd * gcol + becol + x[crow * GN + col];
            if (orow < N) x_new[orow * GN + col] = o;
            s_act[trow][col] = o;
        }
    }
    if (Wi == nullptr) return;                // (uniform)
    __syncthreads();
    // ---- the next layer's node-level products
    load_act(X);
    acc = mma(Wi, Wj, X, zero());
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < N) xa[orow * GN + col] = acc[r];
    }
    acc = mma(Wj, nullptr, X, zero());
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < N) xb[orow * GN + col] = acc[r];
    }
}

}  // namespace

static unsigned g_linear128_mode = 1;   // bit 0: products through the 3-way bf16 split (csplat_linear128_mode); default on
extern "C" int csplat_linear128_mode(unsigned mode) { g_linear128_mode = mode; return 0; }
extern "C" unsigned csplat_linear128_mode_query(void) { return g_linear128_mode; }

extern "C" int csplat_linear128(void *stream, int64_t M, const float *A, const float *W, const float *bias, float alpha, int relu,
                                const float *gather_a, const int64_t *index_a, const float *gather_b, const int64_t *index_b,
                                const float *ln_gamma, const float *ln_beta, float ln_eps, const float *add_pre,
                                const float *add_post, float *out) {
    return csplat_linear128_ex(stream, M, A, W, 128, 0, bias, alpha, relu, gather_a, index_a, gather_b, index_b, ln_gamma, ln_beta, ln_eps,
                               add_pre, add_post, nullptr, nullptr, out);
}

// ---- a narrow first layer: out[M,128] = (ReLU)(x[M,K] W^T + b), K <= 32 (the encoders' first Linear: 4 edge features / 8 node features
// -> 128, /root/reference/meshnet/graph_network.py:48-111).  Paced by the 512 bytes it writes per row; W^T sits in LDS, a thread makes
// 4 adjacent outputs of a row, the 32 threads of a row read the same K inputs (one broadcast line).
namespace {
constexpr int SK_ROWS = 64;               // rows per workgroup pass (8 rows x 8 passes)
template <bool RELU>
__global__ __launch_bounds__(256) void k_linear_narrow(int64_t M, int K, const float *__restrict__ x, int ldx, const float *__restrict__ W,
                                                        int ldw, const float *__restrict__ bias, float *__restrict__ out) {
    __shared__ float4 sWt[32 * 32];       // [k][column group]: W[4g .. 4g + 3][k]
    for (int t = threadIdx.x; t < K * 32; t += 256) {
        const int k = t >> 5, g = t & 31;
        sWt[t] = make_float4(W[(size_t)(4 * g) * ldw + k], W[(size_t)(4 * g + 1) * ldw + k], W[(size_t)(4 * g + 2) * ldw + k], W[(size_t)(4 * g + 3) * ldw + k]);
    }
    __syncthreads();
    const int g = threadIdx.x & 31, r = threadIdx.x >> 5;
    const float4 b = bias ? *reinterpret_cast<const float4 *>(bias + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = (int64_t)blockIdx.x * SK_ROWS; base < M; base += (int64_t)gridDim.x * SK_ROWS) {
#pragma unroll
        for (int p = 0; p < SK_ROWS / 8; p++) {
            const int64_t row = base + 8 * p + r;
            if (row >= M) break;
            const float *xr = x + row * ldx;
            float4 a = b;
            for (int k = 0; k < K; k++) {
                const float v = xr[k];
                const float4 w = sWt[k * 32 + g];
                a.x = fmaf(v, w.x, a.x); a.y = fmaf(v, w.y, a.y); a.z = fmaf(v, w.z, a.z); a.w = fmaf(v, w.w, a.w);
            }
            if (RELU) { a.x = relu_keep_nan(a.x); a.y = relu_keep_nan(a.y); a.z = relu_keep_nan(a.z); a.w = relu_keep_nan(a.w); }
            *reinterpret_cast<float4 *>(out + row * 128 + 4 * g) = a;
        }
    }
}
}  // namespace

extern "C" int csplat_linear_narrow128(void *stream, int64_t M, int K, const float *x, int ldx, const float *W, int ldw, const float *bias,
                                       int relu, float *out) {
    CSPLAT_REQUIRE(M >= 0 && K >= 1 && K <= 32 && ldx >= K && ldw >= K && (M == 0 || (x && W && out)), "csplat_linear_narrow128: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)out | (uintptr_t)bias) & 15u) == 0, "csplat_linear_narrow128: out / bias must be 16-byte aligned");
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    const int64_t nb = (M + SK_ROWS - 1) / SK_ROWS;
    const int grid = (int)(nb < 2048 ? nb : 2048);
    if (relu) k_linear_narrow<true><<<grid, 256, 0, s>>>(M, K, x, ldx, W, ldw, bias, out);
    else k_linear_narrow<false><<<grid, 256, 0, s>>>(M, K, x, ldx, W, ldw, bias, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_linear128_ex(void *stream, int64_t M, const float *A, const float *W, int ldw, int w_transposed, const float *bias,
                                   float alpha, int relu, const float *gather_a, const int64_t *index_a, const float *gather_b,
                                   const int64_t *index_b, const float *ln_gamma, const float *ln_beta, float ln_eps,
                                   const float *add_pre, const float *add_post, const float *mask, float *ln_stats, float *out) {
    CSPLAT_REQUIRE(M >= 0 && (M == 0 || (A && W && out)) && ldw >= 128, "csplat_linear128: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)A | (uintptr_t)out | (uintptr_t)W) & 15u) == 0, "csplat_linear128: A / W / out must be 16-byte aligned");
    const int wt = w_transposed ? 1 : 0;
    if (M == 0) return 0;
    const bool gather = gather_a != nullptr;
    CSPLAT_REQUIRE(!gather || (index_a && gather_b && index_b), "csplat_linear128: gather needs both row sets and both index arrays");
    const bool ln = ln_gamma != nullptr;
    CSPLAT_REQUIRE(!ln || ln_beta, "csplat_linear128: LayerNorm needs gamma and beta");
    const bool add = add_pre != nullptr || add_post != nullptr || mask != nullptr;
    CSPLAT_REQUIRE(!(add && gather), "csplat_linear128: row-aligned addends and gathers are not combined (no caller needs it)");
    CSPLAT_REQUIRE(!(ln && gather), "csplat_linear128: gathers and LayerNorm are not combined (no layer of the network has both)");
    static int s_ok = -1;
    const size_t lds = L128_LDS_F32 + 4 * 64 * sizeof(int);      // W^T + the gather-index strips
    const size_t lds_b3 = L128_LDS_B3 + 8 * 64 * sizeof(int);    // three bf16 pieces of W + strips (one workgroup per CU)
    if (s_ok < 0) {
        s_ok = 1;
        const void *fns[5] = {(const void *)k_linear128<false, false, false, false>, (const void *)k_linear128<false, true, false, false>,
                              (const void *)k_linear128<true, false, false, false>,
                              (const void *)k_linear128<false, false, true, false>,  (const void *)k_linear128<false, true, true, false>};
        for (const void *f : fns) s_ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        const void *fb[5] = {(const void *)k_linear128<false, false, false, true>, (const void *)k_linear128<false, true, false, true>,
                             (const void *)k_linear128<true, false, false, true>,
                             (const void *)k_linear128<false, false, true, true>,  (const void *)k_linear128<false, true, true, true>};
        for (const void *f : fb) s_ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b3) == hipSuccess;
        s_ok &= hipFuncSetAttribute((const void *)k_linear128_rows32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        s_ok &= hipFuncSetAttribute((const void *)k_linear128_rows32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        (void)hipGetLastError();
    }
    CSPLAT_REQUIRE(s_ok, "csplat_linear128: 67 KB of dynamic LDS refused by the runtime");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    const int64_t ntile = (M + 31) / 32;
    if (!gather && ntile <= 2048) {   // node-level sizes: one tile per workgroup, columns split across its waves
        const size_t lds_small = (size_t)(4 * GK * SW_STRIDE + 2 * 32 * 4) * sizeof(float);
        if (ln)
            k_linear128_rows32<true><<<(int)ntile, 256, lds_small, s>>>(M, A, W, bias, alpha, relu, ln_gamma, ln_beta, ln_eps, add_pre, add_post, out, ldw, wt, mask, (float2 *)ln_stats);
        else
            k_linear128_rows32<false><<<(int)ntile, 256, lds_small, s>>>(M, A, W, bias, alpha, relu, ln_gamma, ln_beta, ln_eps, add_pre, add_post, out, ldw, wt, mask, (float2 *)ln_stats);
        LAUNCH_CHECK();
        return 0;
    }
    const bool b3 = (g_linear128_mode & 1u) != 0;
    int grid = b3 ? (int)((ntile + 7) / 8) : (int)((ntile + 3) / 4);
    const int cap = b3 ? 256 : 512;    // persistent: 16 / 2 x 4 wavefronts per CU, weights staged once per workgroup
    if (grid > cap) grid = cap;
#define CSPLAT_L128(G, L, D)                                                                                                  \
    do {                                                                                                                      \
        if (b3)                                                                                                               \
            k_linear128<G, L, D, true><<<grid, 512, lds_b3, s>>>(M, A, W, bias, alpha, relu, gather_a, index_a, gather_b, index_b, \
                                                                 ln_gamma, ln_beta, ln_eps, add_pre, add_post, out, ldw, wt, mask, (float2 *)ln_stats); \
        else                                                                                                                  \
            k_linear128<G, L, D, false><<<grid, 256, lds, s>>>(M, A, W, bias, alpha, relu, gather_a, index_a, gather_b, index_b,  \
                                                               ln_gamma, ln_beta, ln_eps, add_pre, add_post, out, ldw, wt, mask, (float2 *)ln_stats); \
    } while (0)
    if (add && ln) CSPLAT_L128(false, true, true);
    else if (add) CSPLAT_L128(false, false, true);
    else if (gather) CSPLAT_L128(true, false, false);
    else if (ln) CSPLAT_L128(false, true, false);
    else CSPLAT_L128(false, false, false);
#undef CSPLAT_L128
    LAUNCH_CHECK();
    return 0;
}

// (round 3 built a variant of the LayerNorm layer that also summed its rows over the destination nodes in its epilogue -- one more MFMA
//  product with a 0/1 selection matrix + float atomics, csplat_linear128_agg.  Parity-green and no faster -- the LayerNorm layer is bound
//  by its epilogue arithmetic, 110 us with or without it -- it spilled 40 VGPRs and stayed opt-in; it left the library in round 4
//  (history: commit 809fd4b).)

extern "C" int csplat_gnn_node_update(void *stream, int64_t N, const float *agg, const float *x, const float *Wa, const float *Wx,
                                      const float *b0, const float *W2, const float *b2, const float *W3, const float *b3,
                                      const float *ln_gamma, const float *ln_beta, float ln_eps, const float *Wi_next,
                                      const float *Wj_next, float *x_new, float *xa_next, float *xb_next) {
    CSPLAT_REQUIRE(N >= 0 && (N == 0 || (agg && x && Wa && Wx && b0 && W2 && b2 && W3 && b3 && ln_gamma && ln_beta && x_new)),
                   "csplat_gnn_node_update: bad arguments");
    CSPLAT_REQUIRE((Wi_next == nullptr) == (Wj_next == nullptr) && (Wi_next == nullptr || (xa_next && xb_next)),
                   "csplat_gnn_node_update: next-layer weights and outputs come together");
    CSPLAT_REQUIRE(x_new != x && x_new != agg, "csplat_gnn_node_update: x_new must not alias an input (the residual reads x)");
    const uintptr_t al = (uintptr_t)agg | (uintptr_t)x | (uintptr_t)Wa | (uintptr_t)Wx | (uintptr_t)W2 | (uintptr_t)W3 |
                         (uintptr_t)Wi_next | (uintptr_t)Wj_next;
    CSPLAT_REQUIRE((al & 15u) == 0, "csplat_gnn_node_update: operands must be 16-byte aligned");
    if (N == 0) return 0;
    const size_t lds = (size_t)(4 * NU_KH * SW_STRIDE + 32 * ACT_STRIDE + 2 * 32 * 4) * sizeof(float);
    static int s_ok = -1;
    if (s_ok < 0) {
        s_ok = hipFuncSetAttribute((const void *)k_node_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        (void)hipGetLastError();
    }
    CSPLAT_REQUIRE(s_ok, "csplat_gnn_node_update: 52 KB of dynamic LDS refused by the runtime");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    k_node_update<<<(unsigned)((N + 31) / 32), 256, lds, s>>>(N, agg, x, Wa, Wx, b0, W2, b2, W3, b3, ln_gamma, ln_beta, ln_eps,
                                                             Wi_next, Wj_next, x_new, xa_next, xb_next);
    LAUNCH_CHECK();
    return 0;
}

// ---- weight gradient of the 128 -> 128 Linear layers under autograd: dW[o][i] = sum_e g[e][o] * x[e][i]  (g, x [M][128]; M = E
// = 3e5 for the edge MLPs, M = N = 1e4 for the node MLPs).  A [128 x M] x [M x 128] product whose reduction runs over the ROWS:
// as a library call it is reduced inside 16 workgroups (630 us at M = 3e5, 56 us at M = 1e4), as a batched split-K call + sum
// 235 us (round 1).  Here: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) fed from the row-major inputs as they stand.  The MFMA wants
// lane l to supply A[m = l % 32][k = l / 32] and B[k = l / 32][n = l % 32]; k is the ROW of the pair (e, e + 1), and which output
// row a lane's m stands for is ours to choose -- so lane l loads ONE float4 of g and one of x (columns 4c .. 4c + 3 of row e + l / 32,
// c = l % 32: a wave reads two whole 512-byte rows per instruction) and component i of the g vector against component j of the x
// vector is the MFMA for output rows {4m + i} x columns {4n + j}: 16 MFMAs on 16 independent accumulator tiles per row pair, the
// whole 128 x 128 result in one wave's accumulators (256 registers), no LDS, no transposes, every input byte loaded once.
// The 4 waves of a workgroup take different row ranges and are summed through LDS in wave order; per-workgroup partials go to a
// workspace and k_dw128_reduce sums them in workgroup order (deterministic).  Next group's loads are issued before this group's
// MFMAs (one wave per SIMD: the prefetch is what hides HBM latency).  9.8 GFLOP at the 157 TFLOP/s fp32-MFMA rate = 63 us;
// 307 MB at the ~4.5 TB/s a streaming read sustains = 68 us.
namespace {
constexpr int DW_GROUP = 4;        // row pairs per prefetch group
constexpr int DW_WG_MAX = 256;     // one workgroup per CU
constexpr int DW_PART = 4096 + 32;  // float4 per workgroup partial: the 128 x 128 tile + the 128 column sums of g
template <bool BIAS, bool XRELU>
__global__ __launch_bounds__(256, 1) void k_dw128(int64_t M, const float4 *__restrict__ G, const float4 *__restrict__ X,
                                                  float4 *__restrict__ part, int64_t rows_per_wave) {
    __shared__ float4 s_acc[4 * 16 * 64];                               // a quarter of every wave's accumulators: [wave][r][lane]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int kk = lane >> 5, c = lane & 31;
    // the wave's row range, in scalar registers: loop control and the group base addresses stay on the scalar unit
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
    const int64_t r1 = r0 + rows_per_wave < M ? r0 + rows_per_wave : M;
    const int rows = r1 > r0 ? (int)(r1 - r0) : 0;
    const int n_full = rows / (2 * DW_GROUP);                           // groups of DW_GROUP row pairs that need no bounds
    const int total = n_full + (rows % (2 * DW_GROUP) ? 1 : 0);         // + one ragged group
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int lane_at = kk * 32 + c;                                    // float4 index of this lane inside a row pair
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);                      // BIAS: this lane's share of the column sums of g
    auto issue = [&](float4 (&gv)[DW_GROUP], float4 (&xv)[DW_GROUP], int g) {
        const float4 *gp = G + (r0 + (int64_t)g * 2 * DW_GROUP) * 32, *xp = X + (r0 + (int64_t)g * 2 * DW_GROUP) * 32;
#pragma unroll
        for (int u = 0; u < DW_GROUP; u++) { gv[u] = gp[u * 64 + lane_at]; xv[u] = xp[u * 64 + lane_at]; }
    };
    auto issue_ragged = [&](float4 (&gv)[DW_GROUP], float4 (&xv)[DW_GROUP], int g) {
        const float4 *gp = G + (r0 + (int64_t)g * 2 * DW_GROUP) * 32, *xp = X + (r0 + (int64_t)g * 2 * DW_GROUP) * 32;
        const int left = rows - g * 2 * DW_GROUP;
#pragma unroll
        for (int u = 0; u < DW_GROUP; u++) {                            // rows past r1 re-read the last row with a zeroed g operand
            const int e = 2 * u + kk;
            const int at = (e < left ? e : left - 1) * 32 + c;
            const float4 gl = gp[at];
            xv[u] = xp[at];
            gv[u] = e < left ? gl : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto mfma = [&](const float4 (&gv)[DW_GROUP], const float4 (&xv)[DW_GROUP]) {
#pragma unroll
        for (int u = 0; u < DW_GROUP; u++) {
            const float a[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
            float b[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            if (XRELU) {
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = relu_keep_nan(b[j]);
            }
            if (BIAS) { gsum.x += a[0]; gsum.y += a[1]; gsum.z += a[2]; gsum.w += a[3]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // full groups: group g + 1 is in flight under group g's 16 * DW_GROUP MFMAs (the last one re-reads itself: no branch)
    float4 gq[DW_GROUP], xq[DW_GROUP];
    if (n_full > 0) issue(gq, xq, 0);
    for (int g = 0; g < n_full; g++) {
        float4 gn[DW_GROUP], xn[DW_GROUP];
        issue(gn, xn, g + 1 < n_full ? g + 1 : g);
        __builtin_amdgcn_sched_barrier(0);
        mfma(gq, xq);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < DW_GROUP; u++) {
            // the prefetched operands must be in registers HERE: without the pin the compiler is free to sink the loads to their first
            // use at the top of the next iteration (it does in the BIAS variant), i.e. to wait on them right in front of the MFMAs
            asm volatile("" : "+v"(gn[u].x), "+v"(gn[u].y), "+v"(gn[u].z), "+v"(gn[u].w));
            asm volatile("" : "+v"(xn[u].x), "+v"(xn[u].y), "+v"(xn[u].z), "+v"(xn[u].w));
            gq[u] = gn[u]; xq[u] = xn[u];
        }
    }
    if (total > n_full) {                                               // the ragged group, once per wave
        issue_ragged(gq, xq, n_full);
        mfma(gq, xq);
    }
    // the four waves' tiles are summed through LDS a quarter (one i) at a time: every wave parks acc[i][0..3][0..15] as 16 float4 per
    // lane, then wave w sums registers r = w, w + 4, w + 8, w + 12 of the four copies in wave order and stores them (plain LDS reads
    // and writes straight from the accumulator registers: LDS float atomics cost ~200 cycles per wave instruction here).
    // Accumulator register r of lane l holds D[m = 8 * (r / 4) + 4 * (l / 32) + r % 4][n = l % 32] = dW[4 m + i][4 n + j].
    float4 *P = part + (size_t)blockIdx.x * DW_PART;
    if (BIAS) {                                                         // columns 4c .. 4c + 3: the 8 (wave, row parity) shares in order
        s_acc[wave * 64 + lane] = gsum;
        __syncthreads();
        if (threadIdx.x < 32) {
            float4 t = s_acc[threadIdx.x];
#pragma unroll
            for (int k = 1; k < 8; k++) { const float4 o = s_acc[k * 32 + threadIdx.x]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            P[4096 + threadIdx.x] = t;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            s_acc[(wave * 16 + r) * 64 + lane] = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int r = wave + 4 * t;
            float4 v = s_acc[r * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const float4 o = s_acc[(w * 16 + r) * 64 + lane];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            const int m = 8 * (r / 4) + 4 * kk + (r % 4);
            P[(4 * m + i) * 32 + c] = v;
        }
        __syncthreads();
    }
}
// out[t] = sum over partials in workgroup order: blocks x (16 outputs x 16 partial lanes), lanes then combined in lane order;
// float4 t < 4096 is the weight gradient, 4096 .. 4127 the bias gradient
__global__ __launch_bounds__(256) void k_dw128_reduce(int nparts, const float4 *__restrict__ part, float4 *__restrict__ dW,
                                                      float4 *__restrict__ dbias) {
    __shared__ float4 s[256];
    const int o = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + o;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = pl; p < nparts; p += 16) {
        const float4 v = part[(size_t)p * DW_PART + t];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    s[threadIdx.x] = a;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < 16; k++) {
            const float4 v = s[k * 16 + o];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (t < 4096) dW[t] = a; else dbias[t - 4096] = a;
    }
}
// workgroups: one per CU at most, and at least 2 * DW_GROUP row pairs per wave before another workgroup is worth its partial
int dw128_parts(int64_t M) {
    const int64_t want = (M + 4 * 4 * DW_GROUP - 1) / (4 * 4 * DW_GROUP);
    return (int)(want < 1 ? 1 : (want > DW_WG_MAX ? DW_WG_MAX : want));
}
}  // namespace

extern "C" size_t csplat_dw128_workspace_bytes(int64_t M) { return (size_t)dw128_parts(M) * DW_PART * 16; }

extern "C" int csplat_dw128_bias(void *stream, int64_t M, const float *g, const float *x, int x_relu, float *dW, float *dbias, void *workspace) {
    CSPLAT_REQUIRE(M >= 0 && dW && (M == 0 || (g && x && workspace)), "csplat_dw128: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)dW | (uintptr_t)dbias | (uintptr_t)workspace | (uintptr_t)g | (uintptr_t)x) & 15u) == 0, "csplat_dw128: 16-byte aligned buffers");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        HIP_TRY(hipMemsetAsync(dW, 0, 128 * 128 * 4, s));
        if (dbias) HIP_TRY(hipMemsetAsync(dbias, 0, 512, s));
        return 0;
    }
    const int parts = dw128_parts(M);
    int64_t rows = (M + (int64_t)parts * 4 - 1) / ((int64_t)parts * 4);
    rows += rows & 1;                                         // whole row pairs per wave
    const float4 *G = (const float4 *)g, *X = (const float4 *)x;
    float4 *P = (float4 *)workspace;
    if (dbias && x_relu) k_dw128<true, true><<<parts, 256, 0, s>>>(M, G, X, P, rows);
    else if (dbias) k_dw128<true, false><<<parts, 256, 0, s>>>(M, G, X, P, rows);
    else if (x_relu) k_dw128<false, true><<<parts, 256, 0, s>>>(M, G, X, P, rows);
    else k_dw128<false, false><<<parts, 256, 0, s>>>(M, G, X, P, rows);
    LAUNCH_CHECK();
    k_dw128_reduce<<<dbias ? 258 : 256, 256, 0, s>>>(parts, P, (float4 *)dW, (float4 *)dbias);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_dw128(void *stream, int64_t M, const float *g, const float *x, float *dW, void *workspace) {
    return csplat_dw128_bias(stream, M, g, x, 0, dW, nullptr, workspace);
}
