// csplat_edge_mlp.hip -- the WHOLE edge MLP of an InteractionNetwork layer in one launch (rollout / inference, BASELINE configs[3]):
//
//   msg[e] = LayerNorm( W3 relu( W2 relu( alpha * We e0[e] + b0 + xa[dst[e]] + xb[src[e]] ) + b2 ) + b3 )
//
// = /root/reference/meshnet/graph_network.py:178-199 (`message`: edge_fn(cat[x_i, x_j, e]) with the first Linear cut into its three
// column blocks, the x_i / x_j blocks applied at node level: xa = x Wi^T (+ bias rides in b0 here), xb = x Wj^T) for edge features
// alpha * e0 (every layer doubles its edge features, SURVEY F7: alpha = 2^l).
//
// Rounds 1-4 ran this as three csplat_linear128 launches, each a full [E,128] HBM round trip (307 MB per launch at E = 300k: 75-97 us
// each, memory-paced).  Here the two inner [rows,128] activations never leave the registers: algorithmic traffic per layer drops from
// 3 x 307 MB to 154 MB in + 154 MB out (+ the L2-resident gathers), and the kernel is paced by its MFMAs.
//
// Design (gfx950).  One persistent 8-wave workgroup per CU; a wave owns 32 edge rows of a 256-row round and carries them through the
// three layers.  The products run on v_mfma_f32_32x32x16_bf16 with both operands cut into three bf16 pieces (the six partial products
// that matter, fp32 accumulation: fp32-level accuracy at 6/16 of the fp32-MFMA time -- csplat_gemm.hip, B3) and are formed TRANSPOSED:
// MFMA A operand = weight rows (output features), B operand = the lane's edge row.  A lane (row n = lane & 31, half h = lane >> 5)
// then ends a layer holding 64 features of ITS OWN row -- accumulator register r of column tile c <-> feature 32c + 8(r >> 2) + 4h +
// (r & 3) -- which is exactly the B operand of the next layer once that layer's contraction index is permuted to match: step st of the
// next layer contracts over the eight features held in registers 8(st & 1) .. +7 of tile st >> 1.  The permutation is applied to the
// WEIGHTS, once: csplat_gnn_edge_mlp3_pack writes, per layer, the three bf16 pieces of the weight matrix as the exact byte image the
// kernel wants in LDS (rows padded to 272 B: conflict-free 16-byte operand reads), 104,448 B per layer.
// What sank round 2's attempt at this kernel ("k_edge_mlp3": 349 us against 253-276 for three launches) was
// re-staging: cutting fp32 weights into bf16 pieces 3 x per round cost 17 k cycles per layer.  With the image pre-cut, re-staging a
// layer is a straight 102 KiB copy L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 13 wave-instructions per wave, no VGPRs, ~5 k
// cycles with its two barriers); the next round's edge rows are fetched into the registers layer 3's steps free.
// Layout changes (gathered node rows -> lane = edge, finished rows -> whole rows for the stores) are done by the matrix cores
// themselves (identity / selection B operands), so every global access moves whole rows: no LDS scratch, no per-lane row accesses.
//
// MEASURED (round 5, E = 300k, tools/bench_edge_mlp3.py / edge_mlp3_stamps.py / ab_edge_mlp3_rollout.py): parity-green at the first
// run of every version, and NOT faster than the three launches -- 225-270 us against 235-255 us per layer, rollout 6.07 against 5.68 ms
// per step -- so graph_network.EDGE_MLP_FUSED is off by default.  Why, from in-kernel stamps: a round (256 rows) takes ~104 k cycles of
// which the three layers' MFMAs are 37 k: one image in LDS forces the workgroup's 8 waves through the layers in LOCKSTEP (6 barriers
// per round), at 2 waves per SIMD (220 VGPRs) nothing else is resident to fill the gaps, and so every latency is paid in full -- the
// image copies (3 x ~5 k), the barrier skew behind the slower wave of each SIMD (3 x ~5 k), the index -> gather -> cut -> MFMA chains
// of the next round's inputs (~15 k), the LayerNorm / transposition / store issue tail (~25 k).  The three separate launches run 16
// waves per CU out of phase and hide all of it behind 3 x the HBM traffic.  Starting the workgroups staggered changes nothing (the
// phases are not chip-wide bursts on a shared resource).  What would change it: ONE wave per SIMD carrying two 32-row tiles through
// the layers with their memory and MFMA phases interleaved by hand (512 VGPRs: both tiles' rows and accumulators fit, each weight
// operand read serves two MFMAs), i.e. a software-pipelined rewrite of the schedule, not of the data path.
#include "csplat_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int EM_N = 128;                 // layer width
constexpr int EM_STRIDE = 136;            // bf16 elements per image row (272 B)
constexpr int EM_PIECE = EM_N * EM_STRIDE;
constexpr size_t EM_LAYER_BYTES = (size_t)3 * EM_PIECE * 2;      // 104,448 = 102 KiB: three bf16 pieces of one 128 x 128 weight
constexpr int EM_CHUNKS = (int)(EM_LAYER_BYTES / 1024);          // 1 KiB LDS-DMA pieces per layer
static_assert(EM_LAYER_BYTES % 1024 == 0, "the layer image is copied in whole 1 KiB wave-instructions");
constexpr int EM_WAVES = 8, EM_ROWS = 32 * EM_WAVES;             // rows per round
constexpr size_t EM_LDS_BYTES = EM_LAYER_BYTES;

// position pos = 64h + 8st + i of an image row (the element lane-half h feeds into step st as operand element i) <-> source column
__host__ __device__ inline int em_src_col(int layer, int pos) {
    if (layer == 0) return pos;           // layer 1 contracts over the edge row as it lies in memory: half h = columns 64h .. 64h + 63
    const int h = pos >> 6, st = (pos >> 3) & 7, i = pos & 7;
    return 32 * (st >> 1) + 16 * (st & 1) + 8 * (i >> 2) + 4 * h + (i & 3);
}

__global__ __launch_bounds__(EM_STRIDE) void k_edge_mlp3_pack(const float *__restrict__ W0, int ld0, const float *__restrict__ W1, int ld1,
                                                               const float *__restrict__ W2, int ld2, __bf16 *__restrict__ img) {
    const int l = blockIdx.y, j = blockIdx.x, pos = threadIdx.x;
    const float *W = l == 0 ? W0 : (l == 1 ? W1 : W2);
    const int ld = l == 0 ? ld0 : (l == 1 ? ld1 : ld2);
    const float x = pos < EM_N ? W[(size_t)j * ld + em_src_col(l, pos)] : 0.f;      // (pad columns: zeros)
    const __bf16 p1 = (__bf16)x;
    const float r1 = x - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const __bf16 p3 = (__bf16)(r1 - (float)p2);
    __bf16 *row = img + (size_t)l * 3 * EM_PIECE + (size_t)j * EM_STRIDE + pos;
    row[0] = p1; row[EM_PIECE] = p2; row[2 * EM_PIECE] = p3;
}

__device__ __forceinline__ float pair_sum(float v) {      // v of lane (n, 0) + v of lane (n, 1), in both lanes
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

__global__ __launch_bounds__(64 * EM_WAVES) void k_edge_mlp3(int64_t M, const float *__restrict__ e0, float alpha, float inv_alpha,
                                                             const float *__restrict__ xa, const int64_t *__restrict__ ia,
                                                             const float *__restrict__ xb, const int64_t *__restrict__ ib,
                                                             const char *__restrict__ img, const float *__restrict__ b0,
                                                             const float *__restrict__ b1, const float *__restrict__ b2,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                             float *__restrict__ out, int dbg, unsigned long long *__restrict__ stamps) {
    extern __shared__ char s_img[];       // ONE LDS object: the current layer's image
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int64_t nround = (M + EM_ROWS - 1) / EM_ROWS;
    int zs = 0, zv = 0, r32v = r32, hv = h;      // (opaque zeros and the lane ids formed with them per round, see the round loop)
    // (measurement hook, csplat_debug_stamps / tools/edge_mlp3_stamps.py: wave 0 leaves s_memtime at the phase boundaries of its first rounds)
    int stamp_at = 0;
    auto stamp = [&]() {
        if (stamps && w == 0 && stamp_at < 64) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (lane == 0) stamps[(size_t)blockIdx.x * 64 + stamp_at] = t;
            stamp_at++;
        }
    };

    // layer `layer`'s image L2 -> LDS: chunk c (1 KiB) by wave c % 8, lane l moving bytes 16 l .. 16 l + 15 (the LDS image is
    // byte-identical to the global one, padding included, so the lane-linear destination of an LDS-DMA instruction is the layout)
    const unsigned voff = lane * 16;
    bool staged_once = false;
    auto stage = [&](int layer) {
        if ((dbg & 1) && staged_once) return;        // (timing experiment: no re-staging -- results wrong)
        staged_once = true;
        // (the base stays on the scalar unit and is laundered per call: left alone, the compiler forms the 39 per-lane 64-bit source
        //  addresses of the three layers once, outside the round loop, and spills them -- 78 registers)
        const char *sb = img + (size_t)layer * EM_LAYER_BYTES + (size_t)w * 1024;
        asm volatile("" : "+s"(sb));
#pragma unroll
        for (int k = 0; k < (EM_CHUNKS + EM_WAVES - 1) / EM_WAVES; k++) {
            const int c = w + EM_WAVES * k;
            if (c < EM_CHUNKS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + (size_t)k * (EM_WAVES * 1024) + voff),
                                                 (__attribute__((address_space(3))) void *)(s_img + c * 1024), 16, 0, 0);
        }
    };
    // the lane's 64 columns of a [.][128] row, as 16 float4: group g = 4c + q <-> columns 32c + 8q + 4h .. + 3
    auto col_of = [&](int g) { return 32 * (g >> 2) + 8 * (g & 3) + 4 * hv; };

    // one layer's products: acc[c] += W_c (pieces, from LDS) x X (this lane's 64 contraction values, cut into pieces on the fly)
    // refill != nullptr: X[8st .. 8st + 7] is re-loaded from refill[2st], refill[2st + 1] as soon as step st has cut its pieces (the next
    // round's row, fetched into the registers this round frees: no second register set)
    auto products = [&](float (&X)[64], f32x16 (&acc)[4], const float4 *refill) __attribute__((always_inline)) {
        const __bf16 *wl = reinterpret_cast<const __bf16 *>(s_img) + (size_t)r32 * EM_STRIDE + 64 * h;
        auto ldw = [&](int p, int c, int st) {
            return *reinterpret_cast<const bf16x8 *>(wl + ((size_t)p * EM_N + 32 * c) * EM_STRIDE + 8 * st);
        };
        if (dbg & 2) return;                          // (timing experiment: no products)
        bf16x8 wc[3], wn[3];
#pragma unroll
        for (int p = 0; p < 3; p++) wc[p] = ldw(p, 0, 0);
#pragma unroll
        for (int st = 0; st < 8; st++) {
            bf16x8 a1, a2, a3;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = X[8 * st + j];
                a1[j] = (__bf16)x;
                const float r1 = x - (float)a1[j];
                a2[j] = (__bf16)r1;
                a3[j] = (__bf16)(r1 - (float)a2[j]);
            }
            if (refill) {
                const float4 t0 = refill[2 * st], t1 = refill[2 * st + 1];
                X[8 * st] = t0.x; X[8 * st + 1] = t0.y; X[8 * st + 2] = t0.z; X[8 * st + 3] = t0.w;
                X[8 * st + 4] = t1.x; X[8 * st + 5] = t1.y; X[8 * st + 6] = t1.z; X[8 * st + 7] = t1.w;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int cn = (c + 1) & 3, sn = c == 3 ? st + 1 : st;
                if (sn < 8) {
#pragma unroll
                    for (int p = 0; p < 3; p++) wn[p] = ldw(p, cn, sn);
                }
                __builtin_amdgcn_sched_barrier(0);      // next operands requested before this group's 6 MFMAs (192 cycles)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a3, acc[c], 0, 0, 0);      // small terms first
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2], a1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[1], a2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[1], a1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a1, acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; p++) wc[p] = wn[p];
            }
        }
    };
    auto init_bias = [&](const float *__restrict__ b, f32x16 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float4 t = *reinterpret_cast<const float4 *>(b + zs + col_of(g));
            acc[g >> 2][4 * (g & 3)] = t.x; acc[g >> 2][4 * (g & 3) + 1] = t.y; acc[g >> 2][4 * (g & 3) + 2] = t.z; acc[g >> 2][4 * (g & 3) + 3] = t.w;
        }
    };
    // ---- layout changes by MFMA.  Global memory wants whole rows per instruction (lanes along a row: 2 cache lines per half-wave load),
    // the chained layers want lane = edge row (a lane reading its own row's 16 bytes touches 64 lines per instruction, and the
    // texture-address unit prices a memory instruction per line: the gathers, row loads and row stores of the first version took 52 k of a
    // round's 131 k cycles; a transposition through LDS took as long -- tools/edge_mlp3_stamps.py).  The matrix core transposes for free:
    //   gathers   acc[c] (features x edges) += S^T (features x 16 edges, A operand: lane (m, h) element i = S[idx[16kb + 8h + i]][32c + m],
    //             a coalesced dword load per element) x I (16 edges x 32 edges: B operand = the identity block kb)
    //   rows out  O[c] (edges x features) = V (edges x 16 features, A operand = the lane's own finished values, registers 8t .. 8t + 7 of
    //             tile c) x P (16 x 32 selection: feature 32c + j <- the (h, i) that holds it)
    // each with the fp32 operand cut into three bf16 pieces (times exact ones: no rounding beyond the fp32 accumulation).
    auto cut3 = [&](const float (&x)[8], bf16x8 &a1, bf16x8 &a2, bf16x8 &a3) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            a1[j] = (__bf16)x[j];
            const float r1 = x[j] - (float)a1[j];
            a2[j] = (__bf16)r1;
            a3[j] = (__bf16)(r1 - (float)a2[j]);
        }
    };
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto ones_where = [&](auto pred) __attribute__((always_inline)) {      // bf16x8 with 1.0 where pred(i)
        s16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = pred(i) ? (short)0x3F80 : (short)0;
        return __builtin_bit_cast(bf16x8, v);
    };
    auto gather_add = [&](const float *__restrict__ S, int idx, f32x16 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            const bf16x8 eye = ones_where([&](int i) { return r32v == 16 * kb + 8 * hv + i; });
            float v[4][8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int lo = __builtin_amdgcn_readlane(idx, 16 * kb + i), hi = __builtin_amdgcn_readlane(idx, 16 * kb + 8 + i);
                const float *p = S + (size_t)(hv ? hi : lo) * EM_N + r32v;
#pragma unroll
                for (int c = 0; c < 4; c++) v[c][i] = p[32 * c];
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = v[c][i] * inv_alpha;
                bf16x8 a1, a2, a3;
                cut3(x, a1, a2, a3);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, eye, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, eye, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, eye, acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // a round's layer-1 accumulators: (b0 + xa[dst] + xb[src]) / alpha
    auto load_inputs = [&](int64_t round, f32x16 (&acc)[4]) __attribute__((always_inline)) {
        const int64_t row = (round * EM_WAVES + w) * 32 + r32v;
        const int64_t crow = row < M ? row : M - 1;                             // rows past M: clamped loads, masked stores
        const int ja = (int)ia[crow], jb = (int)ib[crow];
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float4 t = *reinterpret_cast<const float4 *>(b0 + zs + col_of(g));
            const int c = g >> 2, r = 4 * (g & 3);
            acc[c][r] = t.x * inv_alpha; acc[c][r + 1] = t.y * inv_alpha; acc[c][r + 2] = t.z * inv_alpha; acc[c][r + 3] = t.w * inv_alpha;
        }
        gather_add(xa, ja, acc);
        gather_add(xb, jb, acc);
    };

    float X[64];                              // the lane's contraction values of the coming layer
    f32x16 acc[4];
    {   // the workgroup's first round: its rows straight into registers (lane = row: 64 lines per instruction, once), its gathers
        const int64_t row = ((int64_t)blockIdx.x * EM_WAVES + w) * 32 + r32;
        const float4 *ap = reinterpret_cast<const float4 *>(e0 + (row < M ? row : M - 1) * EM_N + 64 * h);
#pragma unroll
        for (int q = 0; q < 16; q++) { const float4 t = ap[q]; X[4 * q] = t.x; X[4 * q + 1] = t.y; X[4 * q + 2] = t.z; X[4 * q + 3] = t.w; }
        stage(0);                             // (before the gathers: their waits then cover it -- vmcnt counts in order)
        if (blockIdx.x < nround) load_inputs(blockIdx.x, acc);
    }
    for (int64_t round = blockIdx.x; round < nround; round += gridDim.x) {
        // opaque zeros, renewed per round: everything addressed through them stays INSIDE the loop.  Left alone, the compiler hoists the
        // round-invariant loads (bias, gamma, beta) and per-lane addresses out of the loop and spills them all
        asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
        asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
        r32v = r32 + zv; hv = h + zv;
        stamp();                              // 0: round start -- X, the layer-1 accumulators' start and the image-1 DMA are under way
        // ---------------- layer 1: alpha * We e0 + b0 + xa[dst] + xb[src], ReLU
        asm volatile("s_waitcnt vmcnt(63)" ::: "memory");      // (everything but the previous round's 64 row stores, the youngest, has landed)
        __syncthreads();                      // (every wave's share of the image has landed)
        stamp();                              // 1: layer-1 image in
        products(X, acc, nullptr);
        stamp();                              // 2: layer-1 products done (this wave)
#pragma unroll
        for (int k = 0; k < 64; k++) X[k] = fmaxf(alpha * acc[k >> 4][k & 15], 0.f);
        // ---------------- layer 2
        __syncthreads();
        stamp();                              // 3: every wave's layer-1 products done
        stage(1);
        init_bias(b1, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();                              // 4: layer-2 image in
        products(X, acc, nullptr);
        stamp();                              // 5
#pragma unroll
        for (int k = 0; k < 64; k++) X[k] = fmaxf(acc[k >> 4][k & 15], 0.f);
        // ---------------- layer 3 + LayerNorm
        __syncthreads();
        stamp();                              // 6: every wave's layer-2 products done
        stage(2);
        init_bias(b2, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();                              // 7: layer-3 image in
        // the NEXT round's edge row is fetched during these products, chunk by chunk into the registers the steps free (lane = row: 64
        // cache lines per instruction, which the texture-address unit has all of this layer's MFMAs to work off)
        const bool more = round + gridDim.x < nround;
        {
            const int64_t nrow = ((round + gridDim.x) * EM_WAVES + w) * 32 + r32v;
            const float4 *ap = reinterpret_cast<const float4 *>(e0 + (nrow < M ? nrow : M - 1) * EM_N + 64 * hv);
            products(X, acc, ap);             // (the last round re-reads a clamped row for nothing: no second copy of the loop)
        }
        stamp();                              // 8
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) sum += acc[k >> 4][k & 15];
        const float mean = pair_sum(sum) * (1.f / EM_N);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) { const float d = acc[k >> 4][k & 15] - mean; acc[k >> 4][k & 15] = d; sq += d * d; }
        const float rstd = rsqrtf(pair_sum(sq) * (1.f / EM_N) + eps);
        // edges x features: O[c][r] = the normalised value of edge (r & 3) + 8 (r >> 2) + 4h, feature 32c + r32
        f32x16 O[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < 16; r++) O[c][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const bf16x8 pick = ones_where([&](int i) { return r32v == 16 * t + 8 * (i >> 2) + 4 * hv + (i & 3); });
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = acc[c][8 * t + i] * rstd;
                bf16x8 a1, a2, a3;
                cut3(x, a1, a2, a3);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, pick, O[c], 0, 0, 0);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, pick, O[c], 0, 0, 0);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pick, O[c], 0, 0, 0);
            }
        }
        __syncthreads();                      // (every wave is done with the layer-3 image: the next round's first image may land)
        stamp();                              // 9
        if (more) {
            stage(0);
            load_inputs(round + gridDim.x, acc);
        }
        stamp();                              // 10: next round's inputs requested and in
        {   // whole rows out, LAST: the stores are the youngest memory operations of the wave, so nothing the next round waits for queues
            // behind their drain to HBM (vmcnt counts in order)
            const int64_t base_row = (round * EM_WAVES + w) * 32;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float ga = gamma[zs + 32 * c + r32v], be = beta[zs + 32 * c + r32v];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int64_t orow = base_row + (r & 3) + 8 * (r >> 2) + 4 * hv;
                    if (orow < M) out[orow * EM_N + 32 * c + r32v] = O[c][r] * ga + be;
                }
            }
        }
        stamp();                              // 11: rows out issued
    }
}


// =====================================================================================================================================
// k_edge_mlp3r -- the same message MLP with the WEIGHTS RESIDENT IN REGISTERS (round 5, second design).
//
// What the stamps of k_edge_mlp3 above said: with one weight image in LDS the workgroup re-stages 102 KiB three times per 256 rows and
// walks the layers in lockstep.  Here nothing is re-staged.  One 4-wave workgroup per CU, one wave per SIMD (up to 512 registers each);
// wave j owns output features 32j .. 32j + 31 of ALL THREE layers: 3 layers x 3 bf16 pieces x 8 steps = 72 MFMA A operands = 288
// registers, loaded once per launch.  What moves through LDS is the ACTIVATIONS, already cut into bf16 pieces by whoever produced them:
// a 64-row super-tile (two 32-row tiles = two independent accumulator chains per wave) is [tile][piece][32 rows][128 + 8] bf16, 51 KiB,
// in two buffers the layers ping-pong between.  Per layer a wave issues 2 x 48 MFMAs with B operands read from LDS (one 16-byte read
// per piece and step), turns its 32 x 32 result into the next layer's pieces (ReLU, cut, two 16-byte LDS writes per piece), barrier.
// Global memory is touched in whole rows only: the next super-tile's edge rows and gathered node rows are fetched half-wave-per-row
// into registers while a layer's products run, cut / summed, and parked in LDS (pieces; G = (b0 + xa[dst] + xb[src]) / alpha, which
// layer 1's accumulators START from, as the other layers' start from their bias); LayerNorm's statistics cross the four waves through
// LDS ((sum, M2) per wave, combined by the parallel-variance formula).
// Contraction order: position pos = 64h + 8st + i of an activation row is what lane-half h feeds into step st as element i.  Layer 1:
// pos = column of e0.  Layers 2, 3: the producing wave j' writes its lane's 16 accumulator registers of row n contiguously, pos = 32j'
// + 16h' + r <-> feature 32j' + 8(r >> 2) + 4h' + (r & 3); the permutation is applied to the packed weights (er_src_col).
constexpr int ER_TILE_P = 32 * EM_STRIDE;                 // bf16 elements of one piece of one 32-row tile
constexpr int ER_XBUF = 2 * 3 * ER_TILE_P;                // one activation buffer: [tile 2][piece 3][32][EM_STRIDE]
constexpr int ER_GSTRIDE = 132;                           // floats per G row (528 B: conflict-free 16-byte accesses, lane = row)
constexpr size_t ER_X_BYTES = (size_t)2 * ER_XBUF * 2;    // 104,448
constexpr size_t ER_G_BYTES = (size_t)64 * ER_GSTRIDE * 4;        // 33,792
constexpr size_t ER_S_BYTES = (size_t)2 * 64 * 4 * 8;     // LayerNorm partials [parity][row 64][wave 4] (sum, M2)
constexpr size_t ER_T_BYTES = (size_t)4 * EM_N * 4;       // b1, b2, gamma, beta
constexpr size_t ER_LDS_BYTES = ER_X_BYTES + ER_G_BYTES + ER_S_BYTES + ER_T_BYTES;
constexpr size_t ER_IMAGE_BYTES = (size_t)3 * 4 * 3 * 8 * 64 * 16;      // [layer][wave][piece][step][lane] x 16 B = 294,912

__host__ __device__ inline int er_src_col(int layer, int pos) {
    if (layer == 0) return pos;
    const int j = pos >> 5, h = (pos >> 4) & 1, r = pos & 15;
    return 32 * j + 8 * (r >> 2) + 4 * h + (r & 3);
}

__global__ __launch_bounds__(64) void k_edge_mlp3r_pack(const float *__restrict__ W0, int ld0, const float *__restrict__ W1, int ld1,
                                                        const float *__restrict__ W2, int ld2, bf16x8 *__restrict__ img) {
    // block = (layer l, wave j, step st); lane (m, h): the 8 contraction elements of output feature 32j + m it feeds into step st
    const int st = blockIdx.x & 7, j = (blockIdx.x >> 3) & 3, l = blockIdx.x >> 5;
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const float *W = l == 0 ? W0 : (l == 1 ? W1 : W2);
    const int ld = l == 0 ? ld0 : (l == 1 ? ld1 : ld2);
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float x = W[(size_t)(32 * j + m) * ld + er_src_col(l, 64 * h + 8 * st + i)];
        p1[i] = (__bf16)x;
        const float r1 = x - (float)p1[i];
        p2[i] = (__bf16)r1;
        p3[i] = (__bf16)(r1 - (float)p2[i]);
    }
    bf16x8 *dst = img + ((size_t)((l * 4 + j) * 3) * 8 + st) * 64 + lane;
    dst[0] = p1; dst[8 * 64] = p2; dst[2 * 8 * 64] = p3;
}

__global__ __launch_bounds__(256) void k_edge_mlp3r(int64_t M, const float *__restrict__ e0, float alpha, float inv_alpha,
                                                    const float *__restrict__ xa, const int64_t *__restrict__ ia,
                                                    const float *__restrict__ xb, const int64_t *__restrict__ ib,
                                                    const bf16x8 *__restrict__ wimg, const float *__restrict__ b0,
                                                    const float *__restrict__ b1, const float *__restrict__ b2,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                    float *__restrict__ out, unsigned long long *__restrict__ stamps) {
    extern __shared__ char s_mem[];
    __bf16 *const sX = reinterpret_cast<__bf16 *>(s_mem);
    float *const sG = reinterpret_cast<float *>(s_mem + ER_X_BYTES);
    float2 *const sS = reinterpret_cast<float2 *>(s_mem + ER_X_BYTES + ER_G_BYTES);
    float *const sT = reinterpret_cast<float *>(s_mem + ER_X_BYTES + ER_G_BYTES + ER_S_BYTES);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int64_t nst = (M + 63) / 64;

    int stamp_at = 0;                         // (measurement hook, csplat_debug_stamps / tools/edge_mlp3_stamps.py)
    auto stamp = [&]() {
        if (stamps && w == 0 && stamp_at < 64) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (lane == 0) stamps[(size_t)blockIdx.x * 64 + stamp_at] = t;
            stamp_at++;
        }
    };
    // this wave's 32 output features of the three layers: 72 operands = 288 registers, for the whole launch.  The first 64 operands are
    // PINNED to the accumulation registers (values that only ever meet "a" constraints: the allocator cannot put them anywhere else) and
    // copied next to their use; left to itself the allocator fills the 256 architectural registers with weights first and the loop's own
    // values fight over what is left (operand reads from LDS serialised with their MFMAs: 41 cycles per MFMA instead of 32)
    int Wa[64][4];
    bf16x8 Wv[8];
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int st = 0; st < 8; st++) {
                const int id = (l * 3 + p) * 8 + st;
                const bf16x8 v = wimg[((size_t)((l * 4 + w) * 3 + p) * 8 + st) * 64 + lane];
                if (id < 64) {
                    const i32x4 q = __builtin_bit_cast(i32x4, v);
#pragma unroll
                    for (int c = 0; c < 4; c++) asm("v_accvgpr_write_b32 %0, %1" : "=a"(Wa[id][c]) : "v"(q[c]));
                } else Wv[id - 64] = v;
            }
    int tick = 0;                             // (renewed per layer: an operand copy depends on it, so it stays next to its use)
    auto wop = [&](int l, int p, int st) __attribute__((always_inline)) -> bf16x8 {
        const int id = (l * 3 + p) * 8 + st;
        if (id >= 64) return Wv[id - 64];
        i32x4 q;
#pragma unroll
        for (int c = 0; c < 4; c++) asm("v_accvgpr_read_b32 %0, %1" : "=v"(q[c]) : "a"(Wa[id][c]), "v"(tick));
        return __builtin_bit_cast(bf16x8, q);
    };
    for (int t = threadIdx.x; t < 4 * EM_N; t += 256)
        sT[t] = t < EM_N ? b1[t] : (t < 2 * EM_N ? b2[t - EM_N] : (t < 3 * EM_N ? gamma[t - 2 * EM_N] : beta[t - 3 * EM_N]));
    float4 b0v = *reinterpret_cast<const float4 *>(b0 + 4 * n);       // (loader layout: half-wave per row, lane n <-> columns 4n .. 4n + 3)
    b0v.x *= inv_alpha; b0v.y *= inv_alpha; b0v.z *= inv_alpha; b0v.w *= inv_alpha;

    // ---- loaders.  A wave brings in rows 16w .. 16w + 15 of a super-tile, two rows per instruction (half-wave per row, 16 bytes per lane)
    int iva = 0, ivb = 0;                     // lanes 0 .. 15: the gather indices of the wave's 16 rows
    auto load_idx = [&](int64_t s) {
        int64_t row = s * 64 + 16 * w + (lane & 15);
        row = row < M ? row : M - 1;          // (rows past M: clamped loads, masked stores)
        iva = (int)ia[row]; ivb = (int)ib[row];
    };
    float4 E[8];
    auto issue_e0 = [&](int64_t s) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int64_t row = s * 64 + 16 * w + 2 * k + h;
            row = row < M ? row : M - 1;
            E[k] = *reinterpret_cast<const float4 *>(e0 + row * EM_N + 4 * n);
        }
    };
    auto cut4 = [&](const float4 &v, uint2 &q1, uint2 &q2, uint2 &q3) __attribute__((always_inline)) {
        const float x[4] = {v.x, v.y, v.z, v.w};
        __bf16 p1[4], p2[4], p3[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            p1[t] = (__bf16)x[t];
            const float r1 = x[t] - (float)p1[t];
            p2[t] = (__bf16)r1;
            p3[t] = (__bf16)(r1 - (float)p2[t]);
        }
        q1 = *reinterpret_cast<uint2 *>(p1); q2 = *reinterpret_cast<uint2 *>(p2); q3 = *reinterpret_cast<uint2 *>(p3);
    };
    auto commit_e0 = [&](__bf16 *Xb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int lr = 16 * w + 2 * k + h;                        // row of the super-tile
            __bf16 *dst = Xb + (size_t)(lr >> 5) * 3 * ER_TILE_P + (size_t)(lr & 31) * EM_STRIDE + 4 * n;
            uint2 q1, q2, q3;
            cut4(E[k], q1, q2, q3);
            *reinterpret_cast<uint2 *>(dst) = q1;
            *reinterpret_cast<uint2 *>(dst + ER_TILE_P) = q2;
            *reinterpret_cast<uint2 *>(dst + 2 * ER_TILE_P) = q3;
        }
    };
    float4 GA[4], GB[4];
    auto issue_g = [&](int half) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int kk = 4 * half + k;
            const int a0 = __builtin_amdgcn_readlane(iva, 2 * kk), a1 = __builtin_amdgcn_readlane(iva, 2 * kk + 1);
            const int c0 = __builtin_amdgcn_readlane(ivb, 2 * kk), c1 = __builtin_amdgcn_readlane(ivb, 2 * kk + 1);
            GA[k] = *reinterpret_cast<const float4 *>(xa + (size_t)(h ? a1 : a0) * EM_N + 4 * n);
            GB[k] = *reinterpret_cast<const float4 *>(xb + (size_t)(h ? c1 : c0) * EM_N + 4 * n);
        }
    };
    auto commit_g = [&](int half) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int lr = 16 * w + 2 * (4 * half + k) + h;
            float4 v;
            v.x = (GA[k].x + GB[k].x) * inv_alpha + b0v.x; v.y = (GA[k].y + GB[k].y) * inv_alpha + b0v.y;
            v.z = (GA[k].z + GB[k].z) * inv_alpha + b0v.z; v.w = (GA[k].w + GB[k].w) * inv_alpha + b0v.w;
            *reinterpret_cast<float4 *>(sG + (size_t)lr * ER_GSTRIDE + 4 * n) = v;
        }
    };

    // ---- a layer of both tiles: acc_t += W_l (registers) x X_t (pieces in LDS); the six products that matter, small terms first
    auto products = [&](int l, const __bf16 *Xin, f32x16 &acc0, f32x16 &acc1) __attribute__((always_inline)) {
        const __bf16 *row = Xin + (size_t)n * EM_STRIDE + 64 * h;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tick));
        bf16x8 b[6], bn[6];
#pragma unroll
        for (int p = 0; p < 6; p++) b[p] = *reinterpret_cast<const bf16x8 *>(row + p * ER_TILE_P);
#pragma unroll
        for (int st = 0; st < 8; st++) {
            if (st < 7) {
#pragma unroll
                for (int p = 0; p < 6; p++) bn[p] = *reinterpret_cast<const bf16x8 *>(row + p * ER_TILE_P + 8 * (st + 1));
            }
            const bf16x8 w0 = wop(l, 0, st), w1 = wop(l, 1, st), w2 = wop(l, 2, st);
            __builtin_amdgcn_sched_barrier(0);      // (the next step's operands are requested BEFORE this step's 12 MFMAs, not next to their use)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b[5], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b[3], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b[4], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b[4], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b[3], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b[3], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 6; p++) b[p] = bn[p];
        }
    };
    // the accumulators' start: 16 floats of this lane's features (register r <-> feature 32w + 8(r >> 2) + 4h + (r & 3)) from an LDS row
    auto start_from = [&](const float *base, f32x16 &acc) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 t = *reinterpret_cast<const float4 *>(base + 32 * w + 8 * q + 4 * h);
            acc[4 * q] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w;
        }
    };
    // ReLU(scale * acc) cut into the next layer's pieces: 16 contiguous positions 32w + 16h .. + 15 of row n
    auto relu_to_pieces = [&](const f32x16 &acc, float scale, __bf16 *Xtile) __attribute__((always_inline)) {
        __bf16 p1[16], p2[16], p3[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float x = fmaxf(scale * acc[r], 0.f);
            p1[r] = (__bf16)x;
            const float r1 = x - (float)p1[r];
            p2[r] = (__bf16)r1;
            p3[r] = (__bf16)(r1 - (float)p2[r]);
        }
        __bf16 *dst = Xtile + (size_t)n * EM_STRIDE + 32 * w + 16 * h;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            *reinterpret_cast<uint4 *>(dst + 8 * k) = *reinterpret_cast<uint4 *>(p1 + 8 * k);
            *reinterpret_cast<uint4 *>(dst + ER_TILE_P + 8 * k) = *reinterpret_cast<uint4 *>(p2 + 8 * k);
            *reinterpret_cast<uint4 *>(dst + 2 * ER_TILE_P + 8 * k) = *reinterpret_cast<uint4 *>(p3 + 8 * k);
        }
    };
    // LayerNorm, first half: this wave's (sum, M2 about its own mean) of the row's 32 features it holds
    auto ln_partials = [&](const f32x16 &acc, float2 *srow) __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[r];
        s = pair_sum(s);
        const float mj = s * (1.f / 32.f);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) { const float d = acc[r] - mj; m2 += d * d; }
        m2 = pair_sum(m2);
        if (h == 0) srow[w] = make_float2(s, m2);
    };
    // second half: combine the four waves' partials, normalise, scale / shift, this lane's 4 x 16 bytes of the row out
    auto ln_store = [&](const f32x16 &acc, const float2 *srow, int64_t grow) __attribute__((always_inline)) {
        const float4 u0 = *reinterpret_cast<const float4 *>(srow), u1 = *reinterpret_cast<const float4 *>(srow + 2);
        const float mean = (u0.x + u0.z + u1.x + u1.z) * (1.f / EM_N);
        const float d0 = u0.x * (1.f / 32.f) - mean, d1 = u0.z * (1.f / 32.f) - mean, d2 = u1.x * (1.f / 32.f) - mean, d3 = u1.z * (1.f / 32.f) - mean;
        const float m2 = (u0.y + u0.w + u1.y + u1.w) + 32.f * (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
        const float rstd = rsqrtf(m2 * (1.f / EM_N) + eps);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int f = 32 * w + 8 * q + 4 * h;
            const float4 ga = *reinterpret_cast<const float4 *>(sT + 2 * EM_N + f), be = *reinterpret_cast<const float4 *>(sT + 3 * EM_N + f);
            float4 y;
            y.x = (acc[4 * q] - mean) * rstd * ga.x + be.x; y.y = (acc[4 * q + 1] - mean) * rstd * ga.y + be.y;
            y.z = (acc[4 * q + 2] - mean) * rstd * ga.z + be.z; y.w = (acc[4 * q + 3] - mean) * rstd * ga.w + be.w;
            if (grow < M) *reinterpret_cast<float4 *>(out + grow * EM_N + f) = y;
        }
    };

    int64_t s = blockIdx.x;
    const int64_t stride = gridDim.x;
    // the first super-tile, synchronously
    load_idx(s);
    issue_e0(s);
    issue_g(0); commit_g(0);
    issue_g(1); commit_g(1);
    commit_e0(sX);
    load_idx(s + stride);
    __syncthreads();
    int par = 0;
    for (; s < nst; s += stride, par ^= 1) {
        __bf16 *Xa = sX + (size_t)par * ER_XBUF, *Xb = sX + (size_t)(par ^ 1) * ER_XBUF;
        const int64_t sn = s + stride;
        f32x16 acc0, acc1;
        // ---------------- layer 1: alpha * (We e0 + (b0 + xa[dst] + xb[src]) / alpha), ReLU
        stamp();                              // 0
        start_from(sG + (size_t)n * ER_GSTRIDE, acc0);
        start_from(sG + (size_t)(32 + n) * ER_GSTRIDE, acc1);
        products(0, Xa, acc0, acc1);
        stamp();                              // 1: layer-1 products
        relu_to_pieces(acc0, alpha, Xb);
        relu_to_pieces(acc1, alpha, Xb + 3 * ER_TILE_P);
        stamp();                              // 2: pieces written
        __syncthreads();                      // (X1 complete; every wave has taken its start values out of G)
        stamp();                              // 3: barrier
        // ---------------- layer 2 (the next super-tile's rows and the first half of its gathers travel under it)
        issue_e0(sn);
        issue_g(0);
        start_from(sT, acc0);
        start_from(sT, acc1);
        products(1, Xb, acc0, acc1);
        stamp();                              // 4: loads issued, layer-2 products
        relu_to_pieces(acc0, 1.f, Xa);
        relu_to_pieces(acc1, 1.f, Xa + 3 * ER_TILE_P);
        commit_g(0);
        stamp();                              // 5: pieces + G half written
        __syncthreads();                      // (X2 complete; Xb free)
        stamp();                              // 6: barrier
        // ---------------- layer 3 + LayerNorm
        commit_e0(Xb);
        issue_g(1);
        stamp();                              // 7: next rows cut and parked
        start_from(sT + EM_N, acc0);
        start_from(sT + EM_N, acc1);
        products(2, Xa, acc0, acc1);
        stamp();                              // 8: layer-3 products
        float2 *srow = sS + (size_t)par * 64 * 4;
        ln_partials(acc0, srow + (size_t)n * 4);
        ln_partials(acc1, srow + (size_t)(32 + n) * 4);
        commit_g(1);
        load_idx(sn + stride);
        stamp();                              // 9: partials, G half
        __syncthreads();                      // (partials, next pieces and next G complete)
        stamp();                              // 10: barrier
        ln_store(acc0, srow + (size_t)n * 4, s * 64 + n);
        ln_store(acc1, srow + (size_t)(32 + n) * 4, s * 64 + 32 + n);
    }
}

}  // namespace

// which of the two kernels serves the entry points (development switch, read once): CSPLAT_EM_KERNEL=lds -> k_edge_mlp3 (weights staged
// through LDS), anything else -> k_edge_mlp3r (weights in registers).  The image is laid out for the kernel that will read it.
static bool em_use_regs() {
    static const int v = [] { const char *e = getenv("CSPLAT_EM_KERNEL"); return (e && e[0] == 'l') ? 0 : 1; }();
    return v != 0;
}

extern "C" size_t csplat_gnn_edge_mlp3_image_bytes(void) { return 3 * EM_LAYER_BYTES > ER_IMAGE_BYTES ? 3 * EM_LAYER_BYTES : ER_IMAGE_BYTES; }

extern "C" int csplat_gnn_edge_mlp3_pack(void *stream, const float *W0, int ld0, const float *W1, int ld1, const float *W2, int ld2, void *image) {
    CSPLAT_REQUIRE(W0 && W1 && W2 && image && ld0 >= EM_N && ld1 >= EM_N && ld2 >= EM_N, "csplat_gnn_edge_mlp3_pack: bad arguments");
    CSPLAT_REQUIRE(((uintptr_t)image & 15u) == 0, "csplat_gnn_edge_mlp3_pack: the image must be 16-byte aligned");
    if (em_use_regs()) k_edge_mlp3r_pack<<<3 * 4 * 8, 64, 0, (hipStream_t)stream>>>(W0, ld0, W1, ld1, W2, ld2, (bf16x8 *)image);
    else k_edge_mlp3_pack<<<dim3(EM_N, 3), EM_STRIDE, 0, (hipStream_t)stream>>>(W0, ld0, W1, ld1, W2, ld2, (__bf16 *)image);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_gnn_edge_mlp3(void *stream, int64_t E, const float *e0, float alpha, const float *xa, const int64_t *index_a,
                                    const float *xb, const int64_t *index_b, const void *image, const float *b0, const float *b1,
                                    const float *b2, const float *ln_gamma, const float *ln_beta, float ln_eps, float *out) {
    CSPLAT_REQUIRE(E >= 0 && (E == 0 || (e0 && xa && index_a && xb && index_b && image && b0 && b1 && b2 && ln_gamma && ln_beta && out)),
                   "csplat_gnn_edge_mlp3: bad arguments");
    if (E == 0) return 0;
    const uintptr_t al = (uintptr_t)e0 | (uintptr_t)xa | (uintptr_t)xb | (uintptr_t)image | (uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2 |
                         (uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)out;
    CSPLAT_REQUIRE((al & 15u) == 0, "csplat_gnn_edge_mlp3: operands must be 16-byte aligned");
    CSPLAT_REQUIRE(out != e0, "csplat_gnn_edge_mlp3: out must not alias e0 (rows are re-read by later rounds' prefetch)");
    int ex = 0;
    const float m = frexpf(alpha, &ex);
    CSPLAT_REQUIRE(alpha > 0.f && m == 0.5f, "csplat_gnn_edge_mlp3: alpha must be a power of two (the edge scale 2^l)");
    hipStream_t s = (hipStream_t)stream;
    if (em_use_regs()) {
        static int r_ok = -1;
        if (r_ok < 0) {
            r_ok = hipFuncSetAttribute((const void *)k_edge_mlp3r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ER_LDS_BYTES) == hipSuccess;
            (void)hipGetLastError();
        }
        CSPLAT_REQUIRE(r_ok, "csplat_gnn_edge_mlp3: 141 KB of dynamic LDS refused by the runtime");
        ProfScope ps(PROF_GNN, s);
        const int64_t nst = (E + 63) / 64;
        k_edge_mlp3r<<<(int)(nst < 256 ? nst : 256), 256, ER_LDS_BYTES, s>>>(E, e0, alpha, 1.0f / alpha, xa, index_a, xb, index_b,
                                                                             (const bf16x8 *)image, b0, b1, b2, ln_gamma, ln_beta, ln_eps, out,
                                                                             csplat_stamp_buffer((size_t)256 * 64));
        LAUNCH_CHECK();
        return 0;
    }
    static int s_ok = -1;
    if (s_ok < 0) {
        s_ok = hipFuncSetAttribute((const void *)k_edge_mlp3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EM_LDS_BYTES) == hipSuccess;
        (void)hipGetLastError();
    }
    CSPLAT_REQUIRE(s_ok, "csplat_gnn_edge_mlp3: 102 KB of dynamic LDS refused by the runtime");
    ProfScope ps(PROF_GNN, s);
    const int64_t nround = (E + EM_ROWS - 1) / EM_ROWS;
    const int grid = (int)(nround < 256 ? nround : 256);      // persistent: one 8-wave workgroup per CU
    static const int dbg = getenv("CSPLAT_EM_DEBUG") ? atoi(getenv("CSPLAT_EM_DEBUG")) : 0;      // (timing experiments only)
    k_edge_mlp3<<<grid, 64 * EM_WAVES, EM_LDS_BYTES, s>>>(E, e0, alpha, 1.0f / alpha, xa, index_a, xb, index_b, (const char *)image, b0, b1, b2,
                                                            ln_gamma, ln_beta, ln_eps, out, dbg,
                                                            csplat_stamp_buffer((size_t)grid * 64));
    LAUNCH_CHECK();
    return 0;
}
