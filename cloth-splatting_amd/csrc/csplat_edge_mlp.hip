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
    __bf16 *const sX = reinterpret_cast<__bf16 *>(s_mem);                                     // [slot 2][buffer 2] tiles of three pieces
    float *const sG = reinterpret_cast<float *>(s_mem + ER_X_BYTES);                          // [slot 2][32][ER_GSTRIDE]
    float2 *const sS = reinterpret_cast<float2 *>(s_mem + ER_X_BYTES + ER_G_BYTES);           // [slot 2][32][wave 4] (sum, M2)
    float *const sT = reinterpret_cast<float *>(s_mem + ER_X_BYTES + ER_G_BYTES + ER_S_BYTES);      // b1, b2, gamma, beta
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;

    int stamp_at = 0;                         // (measurement hook, csplat_debug_stamps / tools/edge_mlp3_stamps.py)
    auto stamp = [&]() {
        if (stamps && w == 0 && stamp_at < 64) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (lane == 0) stamps[(size_t)blockIdx.x * 64 + stamp_at] = t;
            stamp_at++;
        }
    };

    // ---- this wave's 32 output features of the three layers: 72 MFMA A operands = 288 registers, for the whole launch.  64 of them
    // are OWNED through "a" constraints (loaded straight into accumulation registers, read there by the MFMAs: the allocator never sees
    // them as something to move), the last 8 live with the loop's own values in the architectural half.
    i32x4 Wa[64];
    bf16x8 Wv[8];
#pragma unroll
    for (int id = 0; id < 72; id++) {
        const int l = id / 24, p = (id / 8) % 3, st = id & 7;
        const bf16x8 *src = wimg + ((size_t)((l * 4 + w) * 3 + p) * 8 + st) * 64 + lane;
        if (id < 64) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(Wa[id]) : "v"(src) : "memory");
        else Wv[id - 64] = *src;
    }
    for (int t = threadIdx.x; t < 4 * EM_N; t += 256)
        sT[t] = t < EM_N ? b1[t] * inv_alpha : (t < 2 * EM_N ? b2[t - EM_N] * inv_alpha : (t < 3 * EM_N ? gamma[t - 2 * EM_N] : beta[t - 3 * EM_N]));
    eps *= inv_alpha * inv_alpha;             // (everything runs divided by alpha: see the side work)
    float4 b0v = *reinterpret_cast<const float4 *>(b0 + 4 * n);       // (loader layout: half-wave per row, lane n <-> columns 4n .. 4n + 3)
    b0v.x *= inv_alpha; b0v.y *= inv_alpha; b0v.z *= inv_alpha; b0v.w *= inv_alpha;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the asm loads are not in the compiler's books)
    __builtin_amdgcn_sched_barrier(0);

    // one product: acc += W[id] x b.  Accumulate chain: an MFMA's D taken whole as the next one's C needs no wait states; B comes from
    // LDS reads (counted by the compiler); the chain's readers run a barrier later
    auto mfma = [&](int id, const bf16x8 &b, f32x16 &acc) __attribute__((always_inline)) {
        if (id < 64) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(Wa[id & 63]), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(Wv[id & 7]), "v"(b));
    };
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto pk = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
        bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
        return __builtin_bit_cast(unsigned, v);
    };
    auto lo_f = [&](unsigned q) { return __uint_as_float(q << 16); };
    auto hi_f = [&](unsigned q) { return __uint_as_float(q & 0xffff0000u); };

    // =================== side work, cut into operations of 1-4 instructions that ride in the gaps between MFMAs.  One wave per SIMD: a gap
    // hides ~5 single-issue instructions, the sixth costs its full price -- so every task is a numbered list of small operations and
    // spread() deals a task's list evenly over a range of a phase's 48 gaps.
    // Everything below works on values scaled by 1 / alpha (a power of two: exact): layer 1 accumulates We e0 + (b0 + xa + xb) / alpha,
    // the biases of layers 2 and 3 are parked divided by alpha, and LayerNorm runs with eps / alpha^2 -- (az - am) / sqrt(a^2 v + eps) =
    // (z - m) / sqrt(v + eps / a^2) -- so no ReLU carries a multiplication.
    // (a macro: the per-gap trip count must be a literal for the loop to unroll before the gap index is known)
#define ER_SPREAD(k, S0, S1, N, OP)                                                                                   \
    do {                                                                                                              \
        if ((k) >= (S0) && (k) < (S1)) {                                                                              \
            const int a_ = ((k) - (S0)) * (N) / ((S1) - (S0)), b_ = ((k) + 1 - (S0)) * (N) / ((S1) - (S0));           \
            _Pragma("unroll") for (int d_ = 0; d_ < ((N) + (S1) - (S0) - 1) / ((S1) - (S0)); d_++) {                  \
                const int m = a_ + d_;                                                                                \
                if (m < b_) { OP; }                                                                                   \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    auto opaque = [&](unsigned &q) __attribute__((always_inline)) { asm volatile("" : "+v"(q)); };      // (keeps a packed pair ONE conversion: see pk)
    // ---- ReLU(acc) -> the next layer's three bf16 pieces, positions 32w + 16h .. + 15 of row n.  62 operations: per value pair
    // (max, max, pack) (low -) (high -) (pack) (low -) (high -) (pack); 3 x 16-byte writes after pairs 0-3 and after pairs 4-7
    constexpr int RELU_OPS = 62;
    float rx0 = 0.f, rx1 = 0.f;
    unsigned P[3][4], rq = 0;
    auto relu_op = [&](int m, const f32x16 &acc, __bf16 *Xtile) __attribute__((always_inline)) {
        const int half = m / 31, mm = m % 31;
        if (mm < 28) {
            const int jj = mm / 7, j = 4 * half + jj, o = mm % 7;
            if (o == 0) { rx0 = fmaxf(acc[2 * j], 0.f); rx1 = fmaxf(acc[2 * j + 1], 0.f); rq = pk(rx0, rx1); opaque(rq); P[0][jj] = rq; }
            else if (o == 1 || o == 4) rx0 -= lo_f(rq);
            else if (o == 2 || o == 5) rx1 -= hi_f(rq);
            else if (o == 3) { rq = pk(rx0, rx1); opaque(rq); P[1][jj] = rq; }
            else P[2][jj] = pk(rx0, rx1);
        } else {
            const int p = mm - 28;
            __bf16 *dst = Xtile + (size_t)p * ER_TILE_P + (size_t)n * EM_STRIDE + 32 * w + 16 * h + 8 * half;
            *reinterpret_cast<uint4 *>(dst) = make_uint4(P[p][0], P[p][1], P[p][2], P[p][3]);
        }
    };
    // ---- LayerNorm, first half: this wave's (sum, M2 about its own mean) of the 32 features it holds of row n: 14 operations
    constexpr int LNP_OPS = 14;
    float ln_s = 0.f, ln_mj = 0.f, ln_m2 = 0.f;
    auto lnp_op = [&](int m, const f32x16 &acc, float2 *srow) __attribute__((always_inline)) {
        if (m < 4) {
            const float t = (acc[4 * m] + acc[4 * m + 1]) + (acc[4 * m + 2] + acc[4 * m + 3]);
            ln_s = m == 0 ? t : ln_s + t;
        } else if (m == 4) {
            ln_s = pair_sum(ln_s);
            ln_mj = ln_s * (1.f / 32.f);
        } else if (m < 13) {
            const int q = m - 5;              // two values each
            float t = q == 0 ? 0.f : ln_m2;
#pragma unroll
            for (int r = 2 * q; r < 2 * q + 2; r++) { const float d = acc[r] - ln_mj; t = fmaf(d, d, t); }
            ln_m2 = t;
        } else {
            ln_m2 = pair_sum(ln_m2);
            srow[w] = make_float2(ln_s, ln_m2);      // (both half-waves write the same pair)
        }
    };
    // ---- second half: the four waves' partials combined (parallel-variance formula), normalise, scale / shift, 4 x 16 bytes out: 18 operations
    constexpr int LNF_OPS = 18;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(M * 512), 0x00020000);      // (rows past M: dropped)
    const int st_lane = n * 512 + (32 * w + 4 * h) * 4;
    float4 lf_u0, lf_u1, lf_ga, lf_be;
    float lf_mean = 0.f, lf_m2 = 0.f, lf_rstd = 0.f, lf_nm = 0.f, lf_y[4];
    auto lnf_op = [&](int m, const f32x16 &acc, const float2 *srow, unsigned tile_off) __attribute__((always_inline)) {
        if (m == 0) {
            lf_u0 = *reinterpret_cast<const float4 *>(srow); lf_u1 = *reinterpret_cast<const float4 *>(srow + 2);
        } else if (m == 1) {
            lf_mean = ((lf_u0.x + lf_u0.z) + (lf_u1.x + lf_u1.z)) * (1.f / EM_N);
        } else if (m == 2) {
            lf_m2 = (lf_u0.y + lf_u0.w) + (lf_u1.y + lf_u1.w);
            const int f = 32 * w + 4 * h;
            lf_ga = *reinterpret_cast<const float4 *>(sT + 2 * EM_N + f); lf_be = *reinterpret_cast<const float4 *>(sT + 3 * EM_N + f);
        } else if (m == 3) {
            const float d0 = fmaf(lf_u0.x, 1.f / 32.f, -lf_mean), d1 = fmaf(lf_u0.z, 1.f / 32.f, -lf_mean);
            lf_y[0] = d0 * d0; lf_y[0] = fmaf(d1, d1, lf_y[0]);
        } else if (m == 4) {
            const float d2 = fmaf(lf_u1.x, 1.f / 32.f, -lf_mean), d3 = fmaf(lf_u1.z, 1.f / 32.f, -lf_mean);
            lf_y[0] = fmaf(d2, d2, lf_y[0]); lf_y[0] = fmaf(d3, d3, lf_y[0]);
        } else if (m == 5) {
            lf_m2 = fmaf(32.f, lf_y[0], lf_m2);
            lf_rstd = __builtin_amdgcn_rsqf(fmaf(lf_m2, 1.f / EM_N, eps));
            lf_nm = -lf_mean * lf_rstd;
        } else {
            const int q = (m - 6) / 3, part = (m - 6) % 3;
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) lf_y[i] = fmaf(acc[4 * q + i], lf_rstd, lf_nm);
            } else if (part == 1) {
                lf_y[0] = fmaf(lf_y[0], lf_ga.x, lf_be.x); lf_y[1] = fmaf(lf_y[1], lf_ga.y, lf_be.y);
                lf_y[2] = fmaf(lf_y[2], lf_ga.z, lf_be.z); lf_y[3] = fmaf(lf_y[3], lf_ga.w, lf_be.w);
                if (q < 3) {
                    const int f = 32 * w + 8 * (q + 1) + 4 * h;
                    lf_ga = *reinterpret_cast<const float4 *>(sT + 2 * EM_N + f); lf_be = *reinterpret_cast<const float4 *>(sT + 3 * EM_N + f);
                }
            } else {
                i32x4 v;
                v[0] = __float_as_int(lf_y[0]); v[1] = __float_as_int(lf_y[1]); v[2] = __float_as_int(lf_y[2]); v[3] = __float_as_int(lf_y[3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, r_out, st_lane + tile_off + 32 * q, 0, 0);
            }
        }
    };
    // ---- loaders.  A wave brings in rows 8w .. 8w + 7 of a tile, two rows per instruction (half-wave per row, 16 bytes per lane): global
    // memory only ever sees whole rows.  Buffer loads: what lies past the last row reads as zero (edge rows) / index 0 (gathers)
    const __amdgpu_buffer_rsrc_t r_e0 = __builtin_amdgcn_make_buffer_rsrc((void *)e0, 0, (int)(M * 512), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ia = __builtin_amdgcn_make_buffer_rsrc((void *)ia, 0, (int)(M * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ib = __builtin_amdgcn_make_buffer_rsrc((void *)ib, 0, (int)(M * 8), 0x00020000);
    const int ld_lane = (8 * w + h) * 512 + 16 * n, ix_lane = (8 * w + h) * 8;
    // the gather indices of the wave's rows, already where the gathers want them: lane (n, h), k <-> row 8w + 2k + h (8 operations)
    auto idx_op = [&](int m, unsigned tile_rows, int (&ja)[4], int (&jb)[4]) __attribute__((always_inline)) {
        const int k = m >> 1;
        if (m & 1) jb[k] = __builtin_amdgcn_raw_buffer_load_b32(r_ib, ix_lane + tile_rows * 8 + 16 * k, 0, 0);
        else ja[k] = __builtin_amdgcn_raw_buffer_load_b32(r_ia, ix_lane + tile_rows * 8 + 16 * k, 0, 0);
    };
    float4 GA[4], GB[4];
    auto g_issue_op = [&](int m, const int (&ja)[4], const int (&jb)[4]) __attribute__((always_inline)) {      // 8 operations
        const int k = m >> 1;
        if (m & 1) GB[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(xb) + (size_t)((unsigned)jb[k] * 512u + 16u * n));
        else GA[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(xa) + (size_t)((unsigned)ja[k] * 512u + 16u * n));
    };
    // G = (xa[dst] + xb[src] + b0) / alpha, what layer 1's accumulators start from: 12 operations
    auto g_commit_op = [&](int m, float *Gt) __attribute__((always_inline)) {
        const int k = m / 3, part = m % 3;
        if (part == 0) { GA[k].x += GB[k].x; GA[k].y += GB[k].y; GA[k].z += GB[k].z; GA[k].w += GB[k].w; }
        else if (part == 1) {
            GA[k].x = fmaf(GA[k].x, inv_alpha, b0v.x); GA[k].y = fmaf(GA[k].y, inv_alpha, b0v.y);
            GA[k].z = fmaf(GA[k].z, inv_alpha, b0v.z); GA[k].w = fmaf(GA[k].w, inv_alpha, b0v.w);
        } else *reinterpret_cast<float4 *>(Gt + (size_t)(8 * w + 2 * k + h) * ER_GSTRIDE + 4 * n) = GA[k];
    };
    auto e_issue_op = [&](int k, float4 (&E)[4], unsigned tile_off) __attribute__((always_inline)) {      // 4 operations
        E[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_e0, ld_lane + tile_off + 1024 * k, 0, 0));
    };
    // an edge row's 16 bytes cut into the three pieces, parked as layer 1's B operand: 4 x 11 operations
    constexpr int EC_OPS = 44;
    unsigned eq0 = 0, eq1 = 0;
    auto e_commit_op = [&](int m, float4 (&E)[4], __bf16 *Xtile) __attribute__((always_inline)) {
        const int k = m / 11, o = m % 11;
        __bf16 *dst = Xtile + (size_t)(8 * w + 2 * k + h) * EM_STRIDE + 4 * n;
        auto pack_out = [&](int p) __attribute__((always_inline)) {
            eq0 = pk(E[k].x, E[k].y); eq1 = pk(E[k].z, E[k].w);
            if (p < 2) { opaque(eq0); opaque(eq1); }
            *reinterpret_cast<uint2 *>(dst + (size_t)p * ER_TILE_P) = make_uint2(eq0, eq1);
        };
        if (o == 0) pack_out(0);
        else if (o == 5) pack_out(1);
        else if (o == 10) pack_out(2);
        else if (o == 1 || o == 6) E[k].x -= lo_f(eq0);
        else if (o == 2 || o == 7) E[k].y -= hi_f(eq0);
        else if (o == 3 || o == 8) E[k].z -= lo_f(eq1);
        else if (o == 4 || o == 9) E[k].w -= hi_f(eq1);
    };

    // =================== a phase: one layer of one tile, 48 MFMAs on one accumulation chain, with side(k) riding behind MFMA k.
    // acc starts from 16 floats of an LDS row (register r <-> feature 32w + 8(r >> 2) + 4h + (r & 3)); the step's three B operands are
    // read one step ahead, the piece the next step needs first first
    auto phase = [&](int l, const __bf16 *Xtile, const float *init, f32x16 &acc, auto &&side) __attribute__((always_inline)) {
        const __bf16 *row = Xtile + (size_t)n * EM_STRIDE + 64 * h;
        bf16x8 bc[3], bn[3];
        bc[2] = *reinterpret_cast<const bf16x8 *>(row + 2 * ER_TILE_P);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 t = *reinterpret_cast<const float4 *>(init + 32 * w + 8 * q + 4 * h);
            acc[4 * q] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w;
        }
        bc[0] = *reinterpret_cast<const bf16x8 *>(row);
        bc[1] = *reinterpret_cast<const bf16x8 *>(row + ER_TILE_P);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 8; st++) {
            // (weight piece, activation piece): the six products that matter, small terms first
            constexpr int WP[6] = {0, 2, 1, 0, 1, 0}, XP[6] = {2, 0, 1, 1, 0, 0};
            constexpr int RD[6] = {-1, 2, -1, 0, -1, 1};      // the next step's piece requested behind MFMA i
#pragma unroll
            for (int i = 0; i < 6; i++) {
                mfma((l * 3 + WP[i]) * 8 + st, bc[XP[i]], acc);
                if (RD[i] >= 0 && st < 7) bn[RD[i]] = *reinterpret_cast<const bf16x8 *>(row + RD[i] * ER_TILE_P + 8 * (st + 1));
#ifndef EM_NOSIDE      // (timing experiment: the bare MFMA pipeline -- results wrong)
                side(6 * st + i);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < 3; p++) bc[p] = bn[p];
        }
    };

    const int T0 = blockIdx.x, stride = gridDim.x, ntiles = (int)((M + 31) / 32);
    __bf16 *const XA = sX, *const XB = sX + 2 * (3 * ER_TILE_P);       // slot s, buffer b: sX + (2s + b) * 3 * ER_TILE_P
    float *const GtA = sG, *const GtB = sG + 32 * ER_GSTRIDE;
    float2 *const SrA = sS + (size_t)n * 4, *const SrB = sS + (size_t)(32 + n) * 4;
    constexpr int XT = 3 * ER_TILE_P;
    int jaA[4], jbA[4], jaB[4], jbB[4];
    float4 EA[4], EB[4];
    {   // the first two tiles' inputs, synchronously
#pragma unroll
        for (int m = 0; m < 8; m++) { idx_op(m, (unsigned)T0 * 32u, jaA, jbA); idx_op(m, (unsigned)(T0 + stride) * 32u, jaB, jbB); }
#pragma unroll
        for (int m = 0; m < 8; m++) g_issue_op(m, jaA, jbA);
#pragma unroll
        for (int k = 0; k < 4; k++) { e_issue_op(k, EA, (unsigned)T0 * 16384u); e_issue_op(k, EB, (unsigned)(T0 + stride) * 16384u); }
#pragma unroll
        for (int m = 0; m < 12; m++) g_commit_op(m, GtA);
#pragma unroll
        for (int m = 0; m < 8; m++) g_issue_op(m, jaB, jbB);
#pragma unroll
        for (int m = 0; m < EC_OPS; m++) e_commit_op(m, EA, XA);      // (one list at a time: the operations of a list share their temporaries)
#pragma unroll
        for (int m = 0; m < EC_OPS; m++) e_commit_op(m, EB, XB);
#pragma unroll
        for (int m = 0; m < 12; m++) g_commit_op(m, GtB);
    }
    f32x16 accA, accB, accLA, accLB;
#pragma unroll
    for (int r = 0; r < 16; r++) accLA[r] = accLB[r] = 0.f;
    unsigned offA_prev = 0xfff00000u, offB_prev = 0xfff00000u;       // (no rows to write yet: past the end of any buffer this kernel takes)
    __syncthreads();
    int x = 0;
    for (int tA = T0; tA < ntiles; tA += 2 * stride, x ^= 1) {
        const int tB = tA + stride, tA2 = tA + 2 * stride, tB2 = tB + 2 * stride;
        __bf16 *const XA0 = XA + x * XT, *const XA1 = XA + (x ^ 1) * XT, *const XB0 = XB + x * XT, *const XB1 = XB + (x ^ 1) * XT;
        stamp();
        // 0: layer 1 of A | LayerNorm partials of the previous B, LayerNorm's end + rows out of the previous A, the next A's indices
        phase(0, XA0, GtA + (size_t)n * ER_GSTRIDE, accA, [&](int k) __attribute__((always_inline)) {
            ER_SPREAD(k, 0, 18, LNP_OPS, lnp_op(m, accLB, SrB));
            ER_SPREAD(k, 14, 44, LNF_OPS, lnf_op(m, accLA, SrA, offA_prev));
            ER_SPREAD(k, 44, 48, 8, idx_op(m, (unsigned)tA2 * 32u, jaA, jbA));
        });
        __syncthreads();
        stamp();
        // 1: layer 1 of B | A's ReLU + pieces, LayerNorm's end + rows out of the previous B, the next B's indices
        phase(0, XB0, GtB + (size_t)n * ER_GSTRIDE, accB, [&](int k) __attribute__((always_inline)) {
            ER_SPREAD(k, 0, 32, LNF_OPS, lnf_op(m, accLB, SrB, offB_prev));
            ER_SPREAD(k, 0, 48, RELU_OPS, relu_op(m, accA, XA1));
            ER_SPREAD(k, 44, 48, 8, idx_op(m, (unsigned)tB2 * 32u, jaB, jbB));
        });
        __syncthreads();
        stamp();
        // 2: layer 2 of A | B's ReLU + pieces, the next A's gathers
        phase(1, XA1, sT, accA, [&](int k) __attribute__((always_inline)) {
            ER_SPREAD(k, 0, 8, 8, g_issue_op(m, jaA, jbA));
            ER_SPREAD(k, 0, 48, RELU_OPS, relu_op(m, accB, XB1));
            ER_SPREAD(k, 32, 48, 12, g_commit_op(m, GtA));
        });
        __syncthreads();
        stamp();
        // 3: layer 2 of B | A's ReLU + pieces, the next B's gathers, the next A's edge rows requested
        phase(1, XB1, sT, accB, [&](int k) __attribute__((always_inline)) {
            ER_SPREAD(k, 0, 8, 8, g_issue_op(m, jaB, jbB));
            ER_SPREAD(k, 8, 12, 4, e_issue_op(m, EA, (unsigned)tA2 * 16384u));
            ER_SPREAD(k, 0, 48, RELU_OPS, relu_op(m, accA, XA0));
            ER_SPREAD(k, 32, 48, 12, g_commit_op(m, GtB));
        });
        __syncthreads();
        stamp();
        // 4: layer 3 of A | B's ReLU + pieces, the next A's edge rows cut and parked, the next B's requested
        phase(2, XA0, sT + EM_N, accLA, [&](int k) __attribute__((always_inline)) {
            ER_SPREAD(k, 0, 4, 4, e_issue_op(m, EB, (unsigned)tB2 * 16384u));
            ER_SPREAD(k, 0, 48, RELU_OPS, relu_op(m, accB, XB0));
            ER_SPREAD(k, 4, 48, EC_OPS, e_commit_op(m, EA, XA1));
        });
        __syncthreads();
        stamp();
        // 5: layer 3 of B | A's LayerNorm partials, the next B's edge rows cut and parked
        phase(2, XB0, sT + EM_N, accLB, [&](int k) __attribute__((always_inline)) {
            ER_SPREAD(k, 0, 20, LNP_OPS, lnp_op(m, accLA, SrA));
            ER_SPREAD(k, 4, 48, EC_OPS, e_commit_op(m, EB, XB1));
        });
        offA_prev = (unsigned)tA * 16384u; offB_prev = (unsigned)tB * 16384u;
        __syncthreads();
    }
    // the pipeline's tail: the last B's partials, both tiles' LayerNorm ends
#pragma unroll
    for (int m = 0; m < LNP_OPS; m++) lnp_op(m, accLB, SrB);
#pragma unroll
    for (int m = 0; m < LNF_OPS; m++) lnf_op(m, accLA, SrA, offA_prev);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < LNF_OPS; m++) lnf_op(m, accLB, SrB, offB_prev);
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
